// mock_kernels.cpp -- TEST INFRASTRUCTURE (tests/test_engine_mock.py): CPU stand-ins for the scoring kernels, so that the REAL
// host side of libddt (csrc/ddt_engine.cpp, ddt_model.cpp, ddt_image.cpp, ddt_choice.cpp, ddt_comm.cpp, ddt_codec.cpp, ddt_sparse_host.cpp) can be built against the
// deferred-execution HIP / RCCL model of mock_runtime.cpp and run without a GPU: model load, image packing, variant choice,
// the feeder's double buffering, the class launches on two streams, the rank-quantised path's workspace slots, the sharded
// jobs -- all under adversarial stream schedules.
//
// A stand-in "launch" enqueues deferred operations that read what the real kernels read -- the packed image the engine
// uploaded (csrc/ddt_internal.h layouts), the tuple lines, for the rank-quantised path the threshold tables and the rank
// workspace (written by a separate "pre-pass" operation, exactly the producer / consumer pair whose ordering the engine is
// responsible for) -- and reduce the leaves in the reference order (RefAcc of ddt_kernels.hip; Core.sv:291-316,486-541), so
// results are checked against the oracle bit for bit.  These functions are NOT the product's compute path and are never
// linked into libddt.so; the GPU parity tests are what holds the real kernels to the oracle.
#include "mock_runtime.cpp"

// what csrc/ddt_checks.cpp bakes into the real library after the instruction-level checks of its device code: the model has no device code,
// so nothing is "unchecked" (a 0 would make the engine's automatic choice avoid the "_s2" / deep kernels' stand-ins)
extern "C" {
extern const int ddt_build_s2_checked = 1;
extern const int ddt_build_dma_checked = 1;
}

#include <algorithm>

namespace ddt {

namespace {

// group g = i / 8 -> ((l0+l1)+(l2+l3))+((l4+l5)+(l6+l7)); acc[g % C] <- s_g + acc[g % C]; total = acc[0] + 0, + acc[1], ...
float reduce(const float* leaf, uint32_t n_trees, uint32_t C, uint32_t sum_mode) {
  if (sum_mode == 1) {
    double d = 0.0;
    for (uint32_t i = 0; i < n_trees; ++i) d += (double)leaf[i];
    return (float)d;
  }
  auto add = [sum_mode](float x, float y) -> float {  // sum_mode 2: the reference adder (ddt_internal.h)
    volatile float v = sum_mode == 2 ? ref_add_exact(x, y) : x + y;
    return v;
  };
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t g = 0; g * 8u < n_trees; ++g) {
    float l[8];
    for (uint32_t u = 0; u < 8; ++u) l[u] = g * 8u + u < n_trees ? leaf[g * 8u + u] : 0.0f;
    const float s = add(add(add(l[0], l[1]), add(l[2], l[3])), add(add(l[4], l[5]), add(l[6], l[7])));
    acc[g % C] = add(s, acc[g % C]);
  }
  float tot = 0.0f;
  for (uint32_t k = 0; k < C; ++k) tot = add(acc[k], tot);
  return tot;
}

inline float f_of(uint32_t b) {
  float f;
  memcpy(&f, &b, 4);
  return f;
}

// tile / stream / generic images: 8-byte records {key, word} in a 1-based heap, leaves behind them, or layout 1 (fused last level)
hipError_t launch_records(const ScoreArgs& args, const Variant& var, hipStream_t s) {
  const ScoreArgs a = args;
  const Variant v = var;
  if (a.ev_mid) (void)hipEventRecord(a.ev_mid, s);
  Op* op = new Op();
  op->cost = (double)a.n * g_cost_row;
  op->run = [=] {
    const uint32_t D = a.levels, W = a.tuple_words, tw = (12u << D) / 4u;
    const uint32_t* img = reinterpret_cast<const uint32_t*>(a.img);
    const bool fused = v.kind == kKindTile && (v.opt & 1);
    const uint32_t feat_off = v.kind == kKindTile ? v.feat_off() : v.kind == kKindStream ? v.feat_off_stream(a.n_trees) : 0u;
    std::vector<float> leaf(a.n_trees);
    for (uint64_t i = 0; i < a.n; ++i) {
      const uint32_t* x = a.tuples + i * W;
      for (uint32_t t = 0; t < a.n_trees; ++t) {
        const uint32_t* tr = img + (size_t)t * tw;
        uint32_t m = 1, lf = 0;
        for (uint32_t lvl = 0; lvl < D; ++lvl) {
          const bool last = fused && lvl == D - 1;
          const uint32_t* rec = last ? tr + (4u << D) / 4u + 4u * (m - (1u << (D - 1))) : tr + 2u * m;
          const uint32_t key = rec[0], word = rec[1], addr = word & 0x7FFFFFFFu;
          const uint32_t j = v.kind == kKindGeneric ? addr : (addr - feat_off) / v.row_bytes();
          const uint32_t raw = x[j], xk = a.ieee ? ieee_key(raw) : raw;
          const uint32_t right = raw == a.miss_raw ? word >> 31 : (uint32_t)!((int32_t)xk < (int32_t)key);
          if (last) lf = rec[2 + right];
          m = 2u * m + right;
        }
        leaf[t] = f_of(fused ? lf : tr[(8u << D) / 4u + m - (1u << D)]);
      }
      a.out[i] = reduce(leaf.data(), a.n_trees, a.clusters, a.sum_mode);
    }
  };
  enqueue(s, op);
  return hipSuccess;
}

// the rank pre-pass of one batch (mock layout of the workspace: q[row][feature])
void enqueue_prepass(const ScoreArgs& a, const Q16Aux& x, hipStream_t s) {
  const uint32_t W = a.tuple_words;
  Op* pre = new Op();
  pre->cost = (double)a.n * g_cost_row * 0.05;
  pre->run = [=] {
    const uint64_t tiles = (a.n + 1023) / 1024;
    for (uint64_t t = 0; t < tiles; ++t) x.tile_flags[t] = 0u;
    for (uint64_t i = 0; i < a.n; ++i)
      for (uint32_t j = 0; j < W; ++j) {
        // (feature compaction, csrc/ddt_choice.cpp: compact column j = column fmap[j] of a row of in_words words; ~0 = padding, reads as 0)
        const uint32_t raw = !x.fmap ? a.tuples[i * W + j] : x.fmap[j] == 0xFFFFFFFFu ? 0u : a.tuples[i * x.in_words + x.fmap[j]];
        uint16_t r;
        if (raw == a.miss_raw) {
          r = 0xFFFFu;
          x.tile_flags[i / 1024] = 1u;
        } else {
          const int32_t key = (int32_t)(a.ieee ? ieee_key(raw) : raw);
          const int32_t* tab = reinterpret_cast<const int32_t*>(x.tables) + (size_t)j * x.Kpad;
          r = (uint16_t)(std::upper_bound(tab, tab + x.tabP[j * 8u], key) - tab);  // keys <= x among the K real ones
        }
        x.q[i * W + j] = r;
      }
  };
  enqueue(s, pre);
}

// rank-quantised path: a PRE-PASS operation writes the feature ranks and the per-tile missing flags into the engine's
// workspace, a SCORING operation reads them (and the fast or the slow image per tile) -- two operations, like the kernels
hipError_t launch_q16(const ScoreArgs& args, const Variant& var, hipStream_t s) {
  const ScoreArgs a = args;
  const Variant v = var;
  const Q16Aux x = *reinterpret_cast<const Q16Aux*>(args.aux);
  const uint32_t W = a.tuple_words;
  if (!x.skip_prepass) enqueue_prepass(a, x, s);
  if (a.ev_mid) (void)hipEventRecord(a.ev_mid, s);
  Op* op = new Op();
  op->cost = (double)a.n * g_cost_row;
  op->run = [=] {
    const uint32_t D = a.levels, half = 1u << D, tw = 2u * half, CT = (uint32_t)v.chunk_trees;
    const uint32_t row = v.wide() ? v.tile() : v.tile() * 2u;  // what a record's row-offset field counts in (wide kernels: half the row's bytes)
    const bool gl = (v.opt & 1) != 0, cm = (v.opt & 4) != 0;
    // "_p" kernels: the image may hold n_segs ensembles (classes) back to back, each with its own cluster-major order
    const uint32_t S = x.n_segs ? x.n_segs : 1u, seg_trees = a.n_trees / S;
    std::vector<float> leaf(seg_trees), img_order(a.n_trees);
    // "_cm" images: PU groups in cluster-major order (csrc/ddt_image.cpp pack_image_q16); original group g sits at position pos[g]
    std::vector<uint32_t> pos(seg_trees / 8u);
    for (uint32_t g = 0; g < (uint32_t)pos.size(); ++g) {
      pos[g] = g;
      if (cm && g < x.real_groups) {
        const uint32_t Cc = a.clusters, c = g % Cc;
        uint32_t start = 0;
        for (uint32_t k = 0; k < c; ++k) start += (x.real_groups + Cc - 1u - k) / Cc;
        pos[g] = start + g / Cc;
      }
    }
    for (uint64_t i = 0; i < a.n; ++i) {
      const bool slow = x.tile_flags[i / 1024] != 0u;
      const uint32_t* img = reinterpret_cast<const uint32_t*>(slow ? x.img_slow : a.img);
      const uint16_t* rk = x.q + i * W;
      for (uint32_t t = 0; t < a.n_trees && v.deep(); ++t) {
        // deep kernels (csrc/ddt_internal.h): K levels over the tree's top records, then one 16-byte record per stage -- pairs carry the
        // node, its two children and the byte offset of the next stage's record block, the terminal stage the last level and its leaves
        const uint32_t K = (uint32_t)v.top, treew = v.tree_bytes_q16() / 4u, topw = (4u << K) / 4u, deepw = v.deep_bytes() / 4u, G = v.deep_stages();
        const uint32_t* top = img + (size_t)(t / CT) * CT * treew + (size_t)(t % CT) * topw;
        const uint32_t* deep = img + (size_t)(t / CT) * CT * treew + (size_t)CT * topw + (size_t)(t % CT) * deepw;
        auto goes_right = [&](uint32_t nd) -> uint32_t {
          const uint32_t off = slow ? ((nd >> 16) & 0xFFFEu) : (nd >> 16);
          const uint32_t f = rk[off / row];
          bool right = f >= (nd & 0xFFFFu);
          if (slow && f == 0xFFFFu) right = ((nd >> 16) & 1u) != 0u;
          return right ? 1u : 0u;
        };
        uint32_t m = 1;
        for (uint32_t lvl = 0; lvl < K; ++lvl) m = 2u * m + goes_right(top[m]);
        uint32_t byte = 16u * (m - (1u << K));
        float lf = 0.0f;
        for (uint32_t g = 0; g < G; ++g) {
          const uint32_t* rec = deep + v.deep_stage_off(g) / 4u + byte / 4u;
          const uint32_t r0 = goes_right(rec[0]);
          if (g + 1u < G) byte = rec[3] + 32u * r0 + 16u * goes_right(rec[1u + r0]);
          else lf = f_of(rec[1u + r0]);
        }
        img_order[t] = lf;
      }
      for (uint32_t t = 0; t < a.n_trees && !v.deep(); ++t) {
        const size_t rec_off = gl ? (size_t)(t / CT) * CT * tw + (size_t)(t % CT) * half : (size_t)t * tw;
        const size_t leaf_off = gl ? (size_t)(t / CT) * CT * tw + (size_t)CT * half + (size_t)(t % CT) * half : (size_t)t * tw + half;
        uint32_t m = 1;
        for (uint32_t lvl = 0; lvl < D; ++lvl) {
          const uint32_t nd = img[rec_off + m];
          const uint32_t off = slow ? ((nd >> 16) & 0xFFFEu) : (nd >> 16);
          const uint32_t f = rk[off / row];
          bool right = f >= (nd & 0xFFFFu);
          if (slow && f == 0xFFFFu) right = ((nd >> 16) & 1u) != 0u;
          m = 2u * m + (right ? 1u : 0u);
        }
        img_order[t] = f_of(img[leaf_off + m - half]);
        // the one-launch multi-class kernel does NOT walk the second sub-group of the chunk the host names (Q16Aux::seg_tail_left: the
        // padding half of a class's partly filled PU group) and adds +0 for its four trees: do exactly that, so that a host that names
        // the wrong chunk drops real trees here as it would on the GPU
        if (S > 1u && x.seg_tail_left && x.seg_chunks && CT == 8u) {
          const uint32_t in_seg = t % seg_trees, chunk = in_seg / CT;
          if (x.seg_chunks - chunk == x.seg_tail_left && in_seg % CT >= 4u) img_order[t] = 0.0f;
        }
      }
      if (x.split > 0u) {
        // a small batch cut into slices of the image (csrc/ddt_kernels.hip score_q16_kernel SPLIT): every slice is a block that starts from a zero
        // accumulator and lets it out -- at its end when the slices are the clusters of a cluster-major image (split_len == 0), at every PU
        // group otherwise (a slice = split_len chunks; the partial sum's position = the group's place in the image); the adds follow in
        // launch_cm_combine
        auto add = [&](float p, float q) -> float {
          volatile float r = a.sum_mode == 2 ? ref_add_exact(p, q) : p + q;
          return r;
        };
        auto group_sum = [&](uint32_t g) -> float {
          const float* l = img_order.data() + g * 8u;
          return add(add(add(l[0], l[1]), add(l[2], l[3])), add(add(l[4], l[5]), add(l[6], l[7])));
        };
        const uint32_t Cc = a.clusters, real = x.real_groups, gpc = std::max(CT / 8u, 1u), real_chunks = (real + gpc - 1u) / gpc;
        uint32_t first = 0;  // chunk (deep kernels: PU group)
        for (uint32_t sl = 0; sl < x.split; ++sl) {
          if (x.split_len && v.deep()) {  // slices of PU groups of THIS launch's image (a part of the ensemble: its groups follow group0 others)
            const uint32_t len = std::min(x.split_len, a.n_trees / 8u - first);
            for (uint32_t g = first; g < first + len; ++g) a.out[(size_t)(x.group0 + g) * x.n_pad + i] = add(group_sum(g), 0.0f);
            first += len;
          } else if (x.split_len) {
            const uint32_t len = std::min(x.split_len, real_chunks - first);
            for (uint32_t g = first * gpc; g < (first + len) * gpc; ++g) a.out[(size_t)g * x.n_pad + i] = add(group_sum(g), 0.0f);
            first += len;
          } else {  // (cm, one group per chunk)
            const uint32_t len = (real + Cc - 1u - sl) / Cc;
            float acc = 0.0f;
            for (uint32_t g = first; g < first + len; ++g) acc = add(group_sum(g), acc);
            a.out[(size_t)sl * x.n_pad + i] = acc;
            first += len;
          }
        }
        continue;
      }
      if (cm && (x.state_in || x.state_out || x.group0)) {
        // one PART of an ensemble scored in parts: the kernel's own accumulate over the groups of this image in cluster-major order,
        // starting from / leaving the sum's state (csrc/ddt_kernels.hip score_q16_kernel, "_cm")
        auto add = [&](float p, float q) -> float {
          volatile float r = a.sum_mode == 2 ? ref_add_exact(p, q) : p + q;
          return r;
        };
        const uint32_t Cc = a.clusters, real = x.real_groups;
        auto size_of = [&](uint32_t c) { return c < Cc ? (real + Cc - 1u - c) / Cc : 0u; };
        uint32_t groups = x.group0, cluster = 0, bound = size_of(0);
        while (cluster < Cc && groups >= bound) bound += size_of(++cluster);
        float acc = x.state_in ? x.state_in[i] : 0.0f, total = x.state_in ? x.state_in[x.n_pad + i] : 0.0f;
        for (uint32_t g = 0; g < a.n_trees / 8u; ++g) {
          const float* l = img_order.data() + g * 8u;
          const float sg = add(add(add(l[0], l[1]), add(l[2], l[3])), add(add(l[4], l[5]), add(l[6], l[7])));
          acc = add(sg, acc);
          if (++groups == bound) {
            total = add(acc, total);
            acc = 0.0f;
            bound += size_of(++cluster);
          }
        }
        if (x.state_out) {
          x.state_out[i] = acc;
          x.state_out[x.n_pad + i] = total;
        } else {
          a.out[i] = total;
        }
        continue;
      }
      float best = 0.0f;
      int32_t arg = 0;
      for (uint32_t sg = 0; sg < S; ++sg) {
        for (uint32_t t = 0; t < seg_trees; ++t) leaf[t] = img_order[sg * seg_trees + pos[t / 8u] * 8u + t % 8u];  // back to the stream order the sum is defined on
        const float v = reduce(leaf.data(), seg_trees, a.clusters, a.sum_mode);
        if (S == 1u) {
          a.out[i] = v;
        } else {
          if (a.out) a.out[(size_t)sg * a.n + i] = v;
          if (sg == 0u || v > best || (best != best && v == v)) {
            best = v;
            arg = (int32_t)sg;
          }
        }
      }
      if (S > 1u && x.labels) x.labels[i] = arg;
    }
  };
  enqueue(s, op);
  return hipSuccess;
}

// "sparse_r_*" (csrc/ddt_sparse_r.hip): the 32-bit rank pre-pass -- the kernel's own search (bucket lookup + probes over the DIRECTORY, then the
// count inside the one block it names) on the tables the engine packed and uploaded; mock layout of the workspace: r[row][feature]
uint32_t rank32_of(const Q16Aux& q, const R32Aux& r, uint32_t j, uint32_t raw, uint32_t ieee) {
  const int32_t key = (int32_t)(ieee ? ieee_key(raw) : raw);
  const uint32_t* P = q.tabP + (size_t)j * 8u;
  const uint32_t Kd = P[0], lo = P[1], shift = P[3], Pp = P[4], koff = P[5], K = P[6], hi_real = P[7], B = 1u << r.blk_log2;
  uint32_t b = ((uint32_t)key - lo) >> shift;
  b = b < kQ16RankBuckets - 1u ? b : kQ16RankBuckets - 1u;
  if (key < (int32_t)lo) b = 0u;
  uint32_t pos = q.tabS[(size_t)j * kQ16RankBuckets + b];
  for (uint32_t step = Pp >> 1; step >= 1u; step >>= 1) {
    uint32_t probe = pos + step - 1u;
    probe = probe < q.Kpad - 1u ? probe : q.Kpad - 1u;
    if ((int32_t)q.tables[(size_t)j * q.Kpad + probe] <= key) pos += step;
  }
  pos = pos < Kd ? pos : Kd;
  uint32_t cnt = 0;
  for (uint32_t i = 0; i < B; ++i) cnt += (int32_t)r.tab[(size_t)koff + (size_t)pos * B + i] <= key ? 1u : 0u;
  uint32_t rk = pos * B + cnt;
  rk = rk < K ? rk : K;
  if (key >= (int32_t)hi_real) rk = K;
  return rk;
}

void enqueue_prepass_r32(const ScoreArgs& a, const SparseAux& x, hipStream_t s) {
  const uint32_t W = a.tuple_words;
  Op* pre = new Op();
  pre->cost = (double)a.n * g_cost_row * 0.05;
  pre->run = [=] {
    const uint32_t T = x.r32.tile;
    for (uint64_t t = 0; t < (a.n + T - 1) / T; ++t) x.q16.tile_flags[t] = 0u;
    for (uint64_t i = 0; i < a.n; ++i)
      for (uint32_t j = 0; j < W; ++j) {
        const uint32_t raw = a.tuples[i * W + j];
        uint32_t out;
        if (raw == a.miss_raw) {
          out = kSrMissing;
          x.q16.tile_flags[i / T] = 1u;
        } else {
          out = (rank32_of(x.q16, x.r32, j, raw, a.ieee) << kSrRankShift) | ((1u << kSrRankShift) - 1u);
        }
        x.r32.r[i * W + j] = out;
      }
  };
  enqueue(s, pre);
}

// ... and the walk over one-word nodes and pair records (csrc/ddt_internal.h "32-bit ranks")
hipError_t launch_sparse_r(const ScoreArgs& args, const Variant& var, hipStream_t s) {
  const ScoreArgs a = args;
  const Variant v = var;
  const SparseAux x = *reinterpret_cast<const SparseAux*>(args.aux);
  if (!x.q16.skip_prepass) enqueue_prepass_r32(a, x, s);
  if (a.ev_mid) (void)hipEventRecord(a.ev_mid, s);
  Op* op = new Op();
  op->cost = (double)a.n * g_cost_row;
  op->run = [=] {
    const uint32_t K = a.levels, W = a.tuple_words, tw = v.top_bytes_sparse() / 4u;
    const uint32_t* top = reinterpret_cast<const uint32_t*>(a.img);
    const uint32_t* deep = reinterpret_cast<const uint32_t*>(x.deep);
    std::vector<float> leaf(a.n_trees);
    for (uint64_t i = 0; i < a.n; ++i) {
      const uint32_t* rk = x.r32.r + i * W;
      const bool slow = x.q16.tile_flags[i / x.r32.tile] != 0u;
      auto right = [&](uint32_t rec) -> uint32_t {
        const uint32_t f = rk[(rec & kSrFeatMask) >> kSrFeatShift];
        if (slow && f == kSrMissing) return (rec & kSrMissRight) ? 1u : 0u;
        return f >= rec ? 1u : 0u;
      };
      for (uint32_t slot = 0; slot < a.n_trees; ++slot) {
        const uint32_t* tr = top + (size_t)slot * tw;
        uint32_t m = 1;
        for (uint32_t lvl = 0; lvl < K; ++lvl) m = 2u * m + right(tr[m]);
        uint32_t byte = tr[0] + 16u * m;
        float lf = 0.0f;
        uint32_t hops = 0;
        for (;;) {
          ++hops;
          const uint32_t* rec = deep + byte / 4u;
          const uint32_t r0 = right(rec[0]), cw = rec[1u + r0];
          if (rec[0] & (r0 ? kSrRightLeaf : kSrLeftLeaf)) {
            lf = f_of(cw);
            break;
          }
          byte = rec[3] + 32u * r0 + 16u * right(cw);
        }
        if (hops > x.max_rounds) lf = __builtin_nanf("");  // the kernel would have stopped gathering: a wrong round count must show
        leaf[slot] = lf;
      }
      if (x.q16.split) {
        // a batch of a few tiles cut into slices of C consecutive PU groups (csrc/ddt_sparse_r.hip score_sparse_r_kernel): every cluster's accumulator
        // takes ONE group's sum per slice and goes out as it is -- out[group][row] = the group's 8-leaf reduce tree + 0; the adds: launch_cm_combine
        auto add = [&](float p, float q) -> float {
          volatile float r = a.sum_mode == 2 ? ref_add_exact(p, q) : p + q;
          return r;
        };
        const uint32_t C = a.clusters ? a.clusters : 1u, G = a.n_trees / 8u;
        for (uint32_t g = 0; g < (G + C - 1u) / C * C; ++g) {
          const float* l = leaf.data() + (size_t)g * 8u;
          a.out[(size_t)g * x.q16.n_pad + i] = g < G ? add(add(add(add(l[0], l[1]), add(l[2], l[3])), add(add(l[4], l[5]), add(l[6], l[7]))), 0.0f) : 0.0f;
        }
        continue;
      }
      a.out[i] = reduce(leaf.data(), a.n_trees, a.clusters, a.sum_mode);
    }
  };
  enqueue(s, op);
  return hipSuccess;
}

// sparse (explicit-children) forests: per PU group of 8 trees the top image (first K levels as a perfect heap, level K-1 as
// 16-byte records), then the deep records -- csrc/ddt_internal.h "Sparse forests", csrc/ddt_sparse.hip
hipError_t launch_sparse(const ScoreArgs& args, const Variant& var, hipStream_t s) {
  const ScoreArgs a = args;
  const Variant v = var;
  const SparseAux x = *reinterpret_cast<const SparseAux*>(args.aux);
  const bool ranked = (v.opt & 1) != 0;  // "sparse_q_*": thresholds are ranks, features come from the pre-pass's workspace
  if (ranked && !x.q16.skip_prepass) enqueue_prepass(a, x.q16, s);
  if (a.ev_mid) (void)hipEventRecord(a.ev_mid, s);
  Op* op = new Op();
  op->cost = (double)a.n * g_cost_row;
  op->run = [=] {
    const uint32_t K = a.levels, W = a.tuple_words, tw = v.top_bytes_sparse() / 4u, feat_off = v.feat_off_sparse(), row = v.row_bytes_sparse();
    const bool dense = (v.opt & 2) != 0;  // "sparse_dk_*": K levels of 8-byte records, level K = deep record at byte 16 m + cbase
    const uint32_t* top = reinterpret_cast<const uint32_t*>(a.img);
    const uint32_t* deep = reinterpret_cast<const uint32_t*>(x.deep);
    std::vector<float> leaf(a.n_trees);
    for (uint64_t i = 0; i < a.n; ++i) {
      const uint32_t* t = a.tuples + i * W;
      const uint16_t* rk = ranked ? x.q16.q + i * W : nullptr;
      auto right = [&](uint32_t key, uint32_t w) -> uint32_t {
        const uint32_t j = ((w & kSpAddrMask) - feat_off) / row;
        if (ranked) return rk[j] == 0xFFFFu ? (uint32_t)((w & kSpMissRight) != 0u) : (uint32_t)(rk[j] >= key);
        const uint32_t raw = t[j], xk = a.ieee ? ieee_key(raw) : raw;
        return raw == a.miss_raw ? (uint32_t)((w & kSpMissRight) != 0u) : (uint32_t)!((int32_t)xk < (int32_t)key);
      };
      for (uint32_t slot = 0; slot < a.n_trees; ++slot) {
        const uint32_t* tr = top + (size_t)slot * tw;
        uint32_t m = 1;
        for (uint32_t lvl = 0; lvl + (dense ? 0u : 1u) < K; ++lvl) m = 2u * m + right(tr[2u * m], tr[2u * m + 1u]);
        const uint32_t mid = (dense && (v.opt & 8)) ? (uint32_t)v.top : 0u;  // "sparse_dm<M>_*": M levels of 8-byte records continue the heap in the deep array
        for (uint32_t j = 0; j < mid; ++j) {
          const uint32_t* r8 = deep + (size_t)((8u * m + tr[0]) / 4u);
          m = 2u * m + right(r8[0], r8[1]);
        }
        const bool pairs = dense && (v.opt & 16) != 0;  // "sparse_dp_*": one 16-byte record {key, left key, right key, three feature numbers + missing directions} for the levels K and K+1
        if (pairs) {
          const uint32_t* pr = deep + (size_t)((16u * m + tr[0]) / 16u) * 4u;
          auto right_j = [&](uint32_t key, uint32_t j, uint32_t miss_right) -> uint32_t {
            if (ranked) return rk[j] == 0xFFFFu ? miss_right : (uint32_t)(rk[j] >= key);
            const uint32_t raw = t[j], xk = a.ieee ? ieee_key(raw) : raw;
            return raw == a.miss_raw ? miss_right : (uint32_t)!((int32_t)xk < (int32_t)key);
          };
          const uint32_t w = pr[3], r0 = right_j(pr[0], w & 0xFFu, (w >> 24) & 1u);
          const uint32_t r1 = right_j(pr[1u + r0], (w >> (8u + 8u * r0)) & 0xFFu, (w >> (25u + r0)) & 1u);
          m = 4u * m + 2u * r0 + r1;
        }
        const uint32_t* rec = !dense ? tr + (4u << K) / 4u + 4u * (m - (1u << (K - 1)))
                              : pairs ? deep + (size_t)((16u * m + tr[0] - (32u << K)) / 16u) * 4u
                              : mid  ? deep + (size_t)((16u * m + tr[0] - (8u << (K + mid))) / 16u) * 4u
                                     : deep + (size_t)((16u * m + tr[0]) / 16u) * 4u;
        for (int guard = 0; guard < 100; ++guard) {
          const uint32_t r = right(rec[0], rec[1]), nxt = rec[2 + r];
          if (rec[1] & (r ? kSpRightLeaf : kSpLeftLeaf)) {
            leaf[slot] = f_of(nxt);
            break;
          }
          rec = deep + (size_t)nxt * 4u;
        }
      }
      a.out[i] = reduce(leaf.data(), a.n_trees, a.clusters, a.sum_mode);
    }
  };
  enqueue(s, op);
  return hipSuccess;
}

// a few of the real table's entries (same names and geometry: csrc/ddt_kernels.hip g_variants); the engine's preference
// lists fall through to what exists
const Variant g_mock_variants[] = {
    Variant{"generic", kKindGeneric, 0, kGenericThreads, 1, 1, 1, 0, 0, &launch_records},
    Variant{"q16_d8_c8_u4_gl_s2_cm_p", kKindQ16, 8, 1024, 1, 8, 4, 1, 15, &launch_q16},
    Variant{"q16_d8_c8_u4_gl_s2_cm_x", kKindQ16, 8, 1024, 1, 8, 4, 1, 7, &launch_q16},
    Variant{"q16_d8_c8_u4_gl_s2_cm", kKindQ16, 8, 1024, 1, 8, 4, 1, 7, &launch_q16},
    Variant{"q16_d8_c8_u4_gl_s2", kKindQ16, 8, 1024, 1, 8, 4, 1, 3, &launch_q16},
    Variant{"q16_d8_c8_u4_gl", kKindQ16, 8, 1024, 1, 8, 4, 1, 1, &launch_q16},
    Variant{"q16_d8_c4_u4", kKindQ16, 8, 1024, 1, 4, 4, 1, 0, &launch_q16},
    Variant{"q16_d6_c16_u4", kKindQ16, 6, 1024, 1, 16, 4, 1, 0, &launch_q16},
    Variant{"q16_d6_c16_u4_s2", kKindQ16, 6, 1024, 1, 16, 4, 1, 2, &launch_q16},   // (images in stream order; the forms that have a cut launch)
    Variant{"q16_d7_c8_u4_s2", kKindQ16, 7, 1024, 1, 8, 4, 1, 2, &launch_q16},
    Variant{"q16_d5_c32_u4_s2", kKindQ16, 5, 1024, 1, 32, 4, 1, 2, &launch_q16},
    Variant{"q16_d4_c64_u8", kKindQ16, 4, 1024, 1, 64, 8, 1, 0, &launch_q16},
    Variant{"q16_d3_c128_u8", kKindQ16, 3, 1024, 1, 128, 8, 1, 0, &launch_q16},
    // deep rank-quantised kernels (opt 4 | 32: cluster-major, deep; last field = K)
    Variant{"q16d_d12_k9_c4_u4_cm", kKindQ16, 12, 1024, 1, 4, 4, 1, 36, &launch_q16, 9},
    Variant{"q16d_d10_k9_c4_u4_cm", kKindQ16, 10, 1024, 1, 4, 4, 1, 36, &launch_q16, 9},
    Variant{"q16d_d11_k8_c8_u4_cm", kKindQ16, 11, 1024, 1, 8, 4, 1, 36, &launch_q16, 8},
    Variant{"q16d_d9_k8_c8_u4_cm", kKindQ16, 9, 1024, 1, 8, 4, 1, 36, &launch_q16, 8},
    Variant{"q16d_d14_k9_c4_u4_cm", kKindQ16, 14, 1024, 1, 4, 4, 1, 36, &launch_q16, 9},
    Variant{"q16d_d15_k8_c8_u4_cm", kKindQ16, 15, 1024, 1, 8, 4, 1, 36, &launch_q16, 8},
    // wide tuples (33..64 words; opt bit 6)
    Variant{"q16w_d8_c8_u4_gl_s2_cm_x", kKindQ16, 8, 1024, 1, 8, 4, 1, 7 | 64, &launch_q16},
    Variant{"q16w_d8_c8_u4_gl", kKindQ16, 8, 1024, 1, 8, 4, 1, 1 | 64, &launch_q16},
    Variant{"q16dw_d12_k9_c4_u4_cm", kKindQ16, 12, 1024, 1, 4, 4, 1, 36 | 64, &launch_q16, 9},
    Variant{"q16dw_d10_k9_c4_u4_cm", kKindQ16, 10, 1024, 1, 4, 4, 1, 36 | 64, &launch_q16, 9},
    Variant{"d8_t1024_r1_c4_u4_dma_f", kKindTile, 8, 1024, 1, 4, 4, 1, 1, &launch_records},
    Variant{"d6_t1024_r1_c16_u4_dma", kKindTile, 6, 1024, 1, 16, 4, 1, 0, &launch_records},
    Variant{"d4_t256_r1_c64_u8_dma", kKindTile, 4, 256, 1, 64, 8, 1, 0, &launch_records},
};

}  // namespace

namespace {
const Variant g_mock_sparse[] = {  // csrc/ddt_sparse.hip DDT_SP(K, U, T)
    Variant{"sparse_q_k8_u8_t1024", kKindSparse, 8, 1024, 1, 8, 8, 1, 1, &launch_sparse},
    Variant{"sparse_q_k6_u8_t1024", kKindSparse, 6, 1024, 1, 8, 8, 1, 1, &launch_sparse},
    Variant{"sparse_k6_u8_t256", kKindSparse, 6, 256, 1, 8, 8, 1, 0, &launch_sparse},
    Variant{"sparse_k8_u8_t512", kKindSparse, 8, 512, 1, 8, 8, 1, 0, &launch_sparse},
    Variant{"sparse_k9_u8_t128", kKindSparse, 9, 128, 1, 8, 8, 1, 0, &launch_sparse},
    Variant{"sparse_dk_k6_u8_t256", kKindSparse, 6, 256, 1, 8, 8, 1, 2, &launch_sparse},
    Variant{"sparse_dk_k8_u8_t256", kKindSparse, 8, 256, 1, 8, 8, 1, 2, &launch_sparse},
    Variant{"sparse_dm1_k8_u8_t256", kKindSparse, 8, 256, 1, 8, 8, 1, 2 | 8, &launch_sparse, 1},
    Variant{"sparse_dm2_k8_u8_t256", kKindSparse, 8, 256, 1, 8, 8, 1, 2 | 8, &launch_sparse, 2},
    Variant{"sparse_dk_k9_u8_t512", kKindSparse, 9, 512, 1, 8, 8, 1, 2, &launch_sparse},
    Variant{"sparse_dp_k8_u8_t256", kKindSparse, 8, 256, 1, 8, 8, 1, 2 | 16, &launch_sparse, 2},
    Variant{"sparse_qd_k8_u8_t1024", kKindSparse, 8, 1024, 1, 8, 8, 1, 3, &launch_sparse},
    Variant{"sparse_qp_k8_u8_t1024", kKindSparse, 8, 1024, 1, 8, 8, 1, 3 | 16, &launch_sparse, 2},
    Variant{"sparse_gf_k6_u8_t256", kKindSparse, 6, 256, 1, 8, 8, 1, 4, &launch_sparse},
    Variant{"sparse_r_k9_u8_t256", kKindSparse, 9, 256, 1, 8, 8, 1, 32, &launch_sparse_r},
    Variant{"sparse_r_k10_u8_t256", kKindSparse, 10, 256, 1, 8, 8, 1, 32, &launch_sparse_r},
    Variant{"sparse_r_k9_u8_t128", kKindSparse, 9, 128, 1, 8, 8, 1, 32, &launch_sparse_r},
};
constexpr int kMockDense = (int)(sizeof(g_mock_variants) / sizeof(g_mock_variants[0]));
}  // namespace
int num_sparse_variants() { return (int)(sizeof(g_mock_sparse) / sizeof(g_mock_sparse[0])); }
const Variant& sparse_variant(int i) { return g_mock_sparse[i]; }
int num_variants() { return kMockDense + num_sparse_variants(); }
const Variant& variant(int i) { return i < kMockDense ? g_mock_variants[i] : sparse_variant(i - kMockDense); }

uint32_t generic_lds_bytes(uint32_t, uint32_t, bool* feat_in_lds, bool* tree_in_lds, uint32_t* top_levels) {
  if (feat_in_lds) *feat_in_lds = false;
  if (tree_in_lds) *tree_in_lds = false;
  if (top_levels) *top_levels = 0;
  return 64u * 1024u;
}
uint32_t stream_blocks_per_cu(uint32_t) { return 1; }
// the synthetic tuple generator (csrc/ddt_kernels.hip synth_tuples_kernel) as a deferred operation on the stream
hipError_t launch_synth_tuples(uint32_t* out, uint64_t row0, size_t n, uint32_t F, int dist, uint32_t missing_bits, hipStream_t s) {
  Op* op = new Op();
  op->run = [=] {
    const uint32_t W = (F + 3u) / 4u * 4u;
    for (size_t r = 0; r < n; ++r)
      for (uint32_t j = 0; j < W; ++j) {
        uint32_t bits = 0u;
        if (j < F) {
          const uint64_t h = splitmix64(kSeedX + (row0 + r) * (uint64_t)F + j);
          float v = (float)(h >> 40) * (1.0f / 16777216.0f);
          if (dist == 1) {
            v = v * 2.0f - 1.0f;
            bits = (((h >> 8) & 0xFFFFull) % 20ull == 0ull) ? missing_bits : f32_bits(v);
          } else {
            bits = f32_bits(v);
          }
        }
        out[r * W + j] = bits;
      }
  };
  enqueue(s, op);
  return hipSuccess;
}

}  // namespace ddt
