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

// Mode flags of CSR 201 per device (EngineCSR.sv:194-205; who reads them: PCIeReceiver.sv:156-176,219-226,241,277,298,
// InputDistributor.sv:113-117,200,213, ResultsCombiner.sv:235-262,350-430):
//   shard_mode DDT_SHARD_TREES  "ensemble spread over all devices, tuples broadcast, partial results aggregated"
//       (DTInference.sv:28-31): every device broadcast_data + aggreg_enabled; trees are split per CSR 203.
//   shard_mode DDT_SHARD_ROWS   "ensemble fits one device, tuples partitioned" (DTInference.sv:33-36): every device
//       broadcast_trees, no aggregation; the host deals batches of CSR201[63:32] lines (4 tuples) round-robin
//       (PCIeReceiver.sv:289-312), results are forwarded, not summed (ResultsCombiner.sv:371-391).
//   device 0 is the host node (receives the PCIe streams), the last device of the list sets last_node so that it
//   stops re-broadcasting (InputDistributor.sv:200,213).
int ddt_csr_encode_ex(const ddt_params* p, uint64_t n_tuples, uint32_t num_devices, uint32_t shard_mode,
                      uint32_t device_index, uint64_t csr[DDT_CSR_COUNT]) {
  if (!p || !csr || num_devices == 0 || num_devices > 20) return DDT_EINVAL;  // device list holds <= 20 ids (CSR 208-210)
  if (shard_mode > DDT_SHARD_ROWS || device_index >= num_devices) return DDT_EINVAL;
  if (p->num_levels < 1 || p->num_levels > 16 || p->num_trees == 0 || p->num_features == 0) return DDT_EINVAL;
  const uint32_t C = p->clusters_per_tuple;
  if (C != 1 && C != 2 && C != 4 && C != 8) return DDT_EINVAL;
  const uint64_t wl = p->weights_lines_per_tree, fl = p->findex_lines_per_tree;
  const uint64_t tl = (p->num_features + 3u) / 4u;
  if (wl > 0xFFFF || fl > 0xFFFF || tl > 0xFFFF) return DDT_EINVAL;
  const uint64_t T = p->num_trees;
  const bool rows = shard_mode == DDT_SHARD_ROWS, multi = num_devices > 1;
  // trees per device: contiguous shards (PCIeReceiver.sv:241-264); the whole ensemble when trees are broadcast
  const uint64_t per_dev = rows ? T : (T + num_devices - 1) / num_devices;
  uint64_t local_w = per_dev * wl - 1, local_f = per_dev * fl;
  if (local_w > 0xFFFF || local_f > 0xFFFF) {  // 16-bit per-device line counters (PCIeReceiver.sv:241-264)
    if (multi && !rows) return DDT_EUNSUPPORTED;
    local_w = local_f = 0xFFFF;  // one receiver for the whole stream (single device / broadcast_trees): fields are don't-care
  }
  const uint64_t groups = (per_dev + 7) / 8, trees_per_pu = (groups + C - 1) / C;  // slots per PU (Core.sv:291-304)
  if (trees_per_pu > 0xFF) return DDT_EUNSUPPORTED;
  memset(csr, 0, sizeof(uint64_t) * DDT_CSR_COUNT);
  csr[0] = 1;  // 200: start
  const bool host = device_index == 0, last = device_index + 1 == num_devices;
  uint64_t flags = 0;                                   // 201
  if (host) flags |= (1ull << 1) | (1ull << 6);         // [1] host_node, [6] pcie_receiver_enabled: the host takes the PCIe streams
  if (multi && !rows) flags |= (1ull << 2) | (1ull << 4);  // [2] broadcast_data, [4] aggreg_enabled
  if (multi && rows) flags |= 1ull << 3;                // [3] broadcast_trees
  if (multi) flags |= 1ull << 5;                        // [5] multiple_nodes
  if (last) flags |= 1ull << 7;                         // [7] last_node
  // [0] data_distributed (every device fed its own tuples over PCIe) is not used by either mode as driven from one host
  csr[1] = flags | ((4 * tl) << 32);                    // [63:32] tuple batch per device: 4 tuples (DTInference.sv:35-36)
  csr[2] = (T * (wl + fl)) | ((T * wl) << 32);                                         // 202
  csr[3] = local_w | (local_f << 16) | ((uint64_t)num_devices << 32);                  // 203
  uint64_t prog = 0;  // every model replica programs cluster (k*C) first; the schedule rotates by one per 8 trees
  for (uint32_t k = 0; k < 8; k += C) prog |= 1ull << k;
  const uint64_t proc = (1ull << C) - 1ull;  // first tuple goes to clusters 0..C-1 (Core.sv:305-316)
  csr[4] = prog | (proc << 8) | (wl << 16) | (fl << 32) | (tl << 48);                  // 204
  csr[5] = (uint64_t)p->missing_bits | ((uint64_t)(p->num_levels & 0xF) << 32) | (trees_per_pu << 36) | ((uint64_t)C << 44);  // 205
  // 206: [7:0] broadcast_address / [15:8] results_address = the ADJACENT device trees/tuples are re-broadcast to and
  // local (or aggregated) results are sent to (SL3TxMux.sv:126,252): the next one of the list, the last one closes the
  // ring at the host; then the SL3 / PCIe packet sizes in lines
  const uint64_t next = last ? 0 : device_index + 1;
  csr[6] = next | (next << 8) | (16ull << 16) | (16ull << 24) | (16ull << 32) | (1ull << 40) | (16ull << 48);
  // 207: result lines this device must see before process_done: all of them on the aggregation chain and on the host,
  // its own share when tuples are partitioned (batches of 4 tuples = one result line, dealt round-robin)
  const uint64_t res_lines = (n_tuples + 3) / 4;
  uint64_t my_lines = res_lines;
  if (rows && multi && !host) my_lines = res_lines / num_devices + (device_index < res_lines % num_devices ? 1 : 0);
  csr[7] = my_lines;
  // 208-210: device ids, 5 bits each at a BYTE stride: devices_list[i] <= data[8*(i%8)+4 : 8*(i%8)] (EngineCSR.sv:250-296);
  // register 210 holds ids 16..19 only
  for (uint32_t d = 0; d < num_devices; ++d) csr[8 + d / 8] |= (uint64_t)(d & 0x1Fu) << (8 * (d % 8));
  return DDT_OK;
}

int ddt_csr_encode(const ddt_params* p, uint64_t n_tuples, uint32_t num_devices, uint64_t csr[DDT_CSR_COUNT]) {
  return ddt_csr_encode_ex(p, n_tuples, num_devices, DDT_SHARD_TREES, 0, csr);
}

int ddt_csr_decode_ex(const uint64_t csr[DDT_CSR_COUNT], ddt_params* p, uint64_t* n_tuples, uint32_t* num_devices,
                      uint32_t* shard_mode, uint32_t* mode_flags, uint8_t device_ids[20]) {
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
  uint32_t nd = (uint32_t)bits(csr[3], 39, 32);
  nd = nd ? nd : 1u;
  if (nd > 20u) return DDT_EINVAL;
  if (num_devices) *num_devices = nd;
  if (shard_mode) *shard_mode = bits(csr[1], 3, 3) ? DDT_SHARD_ROWS : DDT_SHARD_TREES;  // broadcast_trees
  if (mode_flags) *mode_flags = (uint32_t)bits(csr[1], 7, 0);
  if (device_ids)  // the RTL's own slices: devices_list[i] <= data[8*(i%8)+4 : 8*(i%8)] of register 208 + i/8
    for (uint32_t i = 0; i < 20u; ++i) device_ids[i] = (uint8_t)bits(csr[8 + i / 8], 8 * (i % 8) + 4, 8 * (i % 8));
  return DDT_OK;
}

int ddt_csr_decode(const uint64_t csr[DDT_CSR_COUNT], ddt_params* p, uint64_t* n_tuples, uint32_t* num_devices) {
  return ddt_csr_decode_ex(csr, p, n_tuples, num_devices, nullptr, nullptr, nullptr);
}

}  // extern "C"
