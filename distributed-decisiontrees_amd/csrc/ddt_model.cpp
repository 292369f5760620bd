// ddt_model.cpp -- the perfect-tree wire format: validation of the parameter block and parsing of the weights / feature-index streams
// into the trees an engine holds (split out of ddt_engine.cpp in round 6).
//
// Reference interfaces restated here: CSR map rtl/DTEngine/EngineCSR.sv:190-305; stream order and framing
// rtl/DTEngine/PCIeReceiver.sv:136-139,230-312; line packing rtl/DTEngine/core/PipelinedMUX.sv:65; model store
// rtl/DTEngine/core/DTPU.sv:282-354.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "ddt_engine_priv.h"

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
}  // namespace ddt
