// ddt_codec.cpp -- the reference's soft-register (CSR) parameter block <-> ddt_params.
//
// Field positions: rtl/DTEngine/EngineCSR.sv:190-305 (write map).  "minus one" conventions follow the RTL:
// the host writes the plain count for every field except CSR203[15:0] (local weights lines), which the RTL
// expects already decremented ("subtract in SW", EngineCSR.sv:213).
#include <cstring>

#include "../../include/ddt.h"

namespace {
inline uint64_t bits(uint64_t v, int hi, int lo) { return (v >> lo) & ((hi - lo == 63) ? ~0ull : ((1ull << (hi - lo + 1)) - 1ull)); }
inline uint32_t wlines_min(uint32_t D) { return (uint32_t)((((1ull << (D + 1)) - 1) + 3) / 4); }
inline uint32_t flines_min(uint32_t D) { return (uint32_t)((((1ull << D) - 1) + 7) / 8); }
}  // namespace

extern "C" {

int ddt_csr_encode(const ddt_params* p, uint64_t n_tuples, uint32_t num_devices, uint64_t csr[DDT_CSR_COUNT]) {
  if (!p || !csr || num_devices == 0 || num_devices > 20) return DDT_EINVAL;  // device list holds <= 20 ids (CSR 208-210)
  if (p->num_levels < 1 || p->num_levels > 16 || p->num_trees == 0 || p->num_features == 0) return DDT_EINVAL;
  const uint32_t C = p->clusters_per_tuple;
  if (C != 1 && C != 2 && C != 4 && C != 8) return DDT_EINVAL;
  const uint64_t wl = p->weights_lines_per_tree, fl = p->findex_lines_per_tree;
  const uint64_t tl = (p->num_features + 3u) / 4u;
  if (wl > 0xFFFF || fl > 0xFFFF || tl > 0xFFFF) return DDT_EINVAL;
  const uint64_t T = p->num_trees;
  const uint64_t per_dev = (T + num_devices - 1) / num_devices;  // contiguous tree shards (PCIeReceiver.sv:241-264)
  uint64_t local_w = per_dev * wl - 1, local_f = per_dev * fl;
  if (local_w > 0xFFFF || local_f > 0xFFFF) {  // 16-bit per-device line counters (PCIeReceiver.sv:241-264)
    if (num_devices > 1) return DDT_EUNSUPPORTED;
    local_w = local_f = 0xFFFF;  // single device: the receiver never switches device, the fields are don't-care
  }
  const uint64_t groups = (per_dev + 7) / 8, trees_per_pu = (groups + C - 1) / C;  // slots per PU (Core.sv:291-304)
  if (trees_per_pu > 0xFF) return DDT_EUNSUPPORTED;
  memset(csr, 0, sizeof(uint64_t) * DDT_CSR_COUNT);
  csr[0] = 1;  // 200: start
  const bool multi = num_devices > 1;
  // 201: [1] host_node [2] broadcast_data [4] aggreg_enabled [5] multiple_nodes [6] pcie_receiver_enabled
  //      [7] last_node (single device: host is also last), [63:32] tuple batch per device in lines
  csr[1] = (1ull << 1) | (1ull << 6) | (multi ? ((1ull << 2) | (1ull << 4) | (1ull << 5)) : (1ull << 7)) | ((4 * tl) << 32);
  csr[2] = (T * (wl + fl)) | ((T * wl) << 32);                                         // 202
  csr[3] = local_w | (local_f << 16) | ((uint64_t)num_devices << 32);                  // 203
  uint64_t prog = 0;  // every model replica programs cluster (k*C) first; the schedule rotates by one per 8 trees
  for (uint32_t k = 0; k < 8; k += C) prog |= 1ull << k;
  const uint64_t proc = (1ull << C) - 1ull;  // first tuple goes to clusters 0..C-1 (Core.sv:305-316)
  csr[4] = prog | (proc << 8) | (wl << 16) | (fl << 32) | (tl << 48);                  // 204
  csr[5] = (uint64_t)p->missing_bits | ((uint64_t)(p->num_levels & 0xF) << 32) | (trees_per_pu << 36) | ((uint64_t)C << 44);  // 205
  csr[6] = (1ull << 0) | (0ull << 8) | (16ull << 16) | (16ull << 24) | (16ull << 32) | (1ull << 40) | (16ull << 48);  // 206
  csr[7] = (n_tuples + 3) / 4;                                                         // 207: result lines
  for (uint32_t d = 0; d < num_devices; ++d) csr[8 + d / 8] |= (uint64_t)d << (5 * (d % 8));  // 208-210: 5-bit ids
  return DDT_OK;
}

int ddt_csr_decode(const uint64_t csr[DDT_CSR_COUNT], ddt_params* p, uint64_t* n_tuples, uint32_t* num_devices) {
  if (!csr || !p) return DDT_EINVAL;
  memset(p, 0, sizeof(*p));
  const uint64_t total_lines = bits(csr[2], 31, 0), weight_lines = bits(csr[2], 63, 32);
  const uint64_t wl = bits(csr[4], 31, 16), fl = bits(csr[4], 47, 32), tl = bits(csr[4], 63, 48);
  uint32_t D = (uint32_t)bits(csr[5], 35, 32);
  if (D == 0) D = 16;  // the RTL keeps num_levels - 1 in 4 bits: 16 levels are written as 0 (EngineCSR.sv:230)
  const uint32_t C = (uint32_t)bits(csr[5], 47, 44);
  if (wl == 0 || fl == 0 || tl == 0 || weight_lines == 0 || weight_lines % wl) return DDT_EINVAL;
  const uint64_t T = weight_lines / wl;
  if (total_lines != T * (wl + fl)) return DDT_EINVAL;
  if (C != 1 && C != 2 && C != 4 && C != 8) return DDT_EINVAL;
  if (wl < wlines_min(D) || fl < flines_min(D) || T > 0xFFFFFFFFull || tl * 4 > 2048) return DDT_EINVAL;
  p->num_trees = (uint32_t)T;
  p->num_levels = D;
  p->num_features = (uint32_t)(tl * 4);
  p->missing_bits = (uint32_t)bits(csr[5], 31, 0);
  p->weights_lines_per_tree = (uint32_t)wl;
  p->findex_lines_per_tree = (uint32_t)fl;
  p->clusters_per_tuple = C;
  if (n_tuples) *n_tuples = bits(csr[7], 31, 0) * 4ull;
  if (num_devices) {
    const uint32_t nd = (uint32_t)bits(csr[3], 39, 32);
    *num_devices = nd ? nd : 1u;
  }
  return DDT_OK;
}

}  // extern "C"
