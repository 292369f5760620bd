// ddt_cli -- stand-in for the reference's (unpublished) host program.
//
// It consumes exactly what a host would hand the FPGA engine: the soft-register block (CSR 200..211,
// rtl/DTEngine/EngineCSR.sv:190-305) and the three inbound line streams in the reference wire format
// (rtl/DTEngine/PCIeReceiver.sv:136-139: all weights lines, then all feature-index lines, then tuple lines;
// word packing rtl/DTEngine/core/PipelinedMUX.sv:65), and writes the outbound result-line stream
// (rtl/DTEngine/ResultsCombiner.sv:136-160: four fp32 scores per 128-bit line, tuple order).
//
//   ddt_cli gen   --trees T --levels D --features F --rows N [--dist 0|1] [--devices G] --prefix DIR/name
//        writes name.csr (text: "<addr> <hex64>" per register), name.weights, name.findex, name.tuples
//   ddt_cli score --csr f.csr --weights f.weights --findex f.findex --tuples f.tuples --out f.results
//                 [--device 0] [--shard i --of n] [--cmp-mode 0|1] [--sum-mode 0|1] [--variant v]
//                 [--devices G [--combine allreduce|chain]]   the whole multi-GPU job in this process: tree shard g on
//                                                             device g, partial scores combined over RCCL (ddt_group_*)
//                 [--devices G --mode rows]                   the reference's other mode: the whole ensemble on every device, the
//                                                             tuples partitioned, every device feeds itself; no collective
//                 [--devices G --mode hybrid --tree-ranks Gt] the two composed (ddt_group_create_hybrid): row groups of Gt consecutive
//                                                             devices; device i holds tree shard i % Gt, a row group scores its slice
//                                                             of the rows, partial scores combined inside the row group
//                 [--ranks N --rank r --id-file PATH [--combine allreduce|chain] [--device d]]   one process per GPU: this
//                                                             process is rank r (tree shard r, device r unless --device), the RCCL
//                                                             id is published by rank 0 through PATH (fresh per job); every
//                                                             rank gets all scores, --out is optional (ddt_comm_*);
//                                                             + --mode hybrid --tree-ranks Gt: rank r holds tree shard r % Gt of Gt
//                                                             (ddt_comm_create_hybrid: the row groups' communicators are split off
//                                                             the one id), host tuples cross PCIe once in the whole job
//   ddt_cli gen-sparse   --trees T --max-depth D --features F --rows N [--full-levels L] [--permille P] [--dist 0|1] --prefix DIR/name
//        a random-forest-like SPARSE model (include/ddt.h ddt_load_model_sparse): name.nodes (one 128-bit line per
//        internal node), name.first (u64 line index of every tree's root, T + 1 entries), name.tuples
//   ddt_cli score-sparse --nodes f.nodes --first f.first --tuples f.tuples --features F --max-depth D --out f.results
//                        [--device 0] [--shard i --of n] [--cmp-mode 0|1] [--sum-mode 0|1] [--clusters C] [--missing 0x7FC00000]
//   ddt_cli info
//
// All scoring goes through the C-ABI of include/ddt.h; there is no CPU fallback.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <vector>

#include "../../include/ddt.h"

namespace {

// a model that landed on a correctness kernel (ddt_info::fallback_kernel): never silently
void warn_fallback(const ddt_info& info) {
  if (info.fallback_kernel)
    fprintf(stderr, "ddt_cli: WARNING: this model runs on the fallback kernel '%s' (no tuned kernel for depth %u / %u features): results are exact, "
                    "throughput is far below the tuned paths\n", info.variant_name, info.num_levels, info.num_features);
}

std::map<std::string, std::string> parse(int argc, char** argv, int first) {
  std::map<std::string, std::string> o;
  for (int i = first; i + 1 < argc; i += 2) {
    if (strncmp(argv[i], "--", 2)) {
      fprintf(stderr, "bad option %s\n", argv[i]);
      exit(2);
    }
    o[argv[i] + 2] = argv[i + 1];
  }
  return o;
}

uint64_t num(const std::map<std::string, std::string>& o, const char* k, uint64_t dflt, bool required = false) {
  auto it = o.find(k);
  if (it == o.end()) {
    if (required) {
      fprintf(stderr, "missing --%s\n", k);
      exit(2);
    }
    return dflt;
  }
  return strtoull(it->second.c_str(), nullptr, 0);
}

std::string str(const std::map<std::string, std::string>& o, const char* k) {
  auto it = o.find(k);
  if (it == o.end()) {
    fprintf(stderr, "missing --%s\n", k);
    exit(2);
  }
  return it->second;
}

bool read_file(const std::string& path, std::vector<unsigned char>* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out->resize((size_t)n);
  const bool ok = n == 0 || fread(out->data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}

bool write_file(const std::string& path, const void* p, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = n == 0 || fwrite(p, 1, n, f) == n;
  fclose(f);
  return ok;
}

bool read_csr(const std::string& path, uint64_t csr[DDT_CSR_COUNT]) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  memset(csr, 0, sizeof(uint64_t) * DDT_CSR_COUNT);
  char line[256];
  while (fgets(line, sizeof(line), f)) {
    if (line[0] == '#' || line[0] == '\n') continue;
    unsigned addr;
    unsigned long long val;
    if (sscanf(line, "%u %llx", &addr, &val) == 2 && addr >= DDT_CSR_FIRST && addr < DDT_CSR_FIRST + DDT_CSR_COUNT)
      csr[addr - DDT_CSR_FIRST] = val;
  }
  fclose(f);
  return true;
}

// The communicator id of a one-process-per-GPU job travels through a file: rank 0 makes it (ddt_comm_get_unique_id) and
// publishes it with an atomic rename, the other ranks wait for the file (any launcher-side channel would do: include/ddt.h).
bool exchange_id(const std::string& path, int rank, int timeout_s, unsigned char id[DDT_COMM_ID_BYTES]) {
  if (rank == 0) {
    if (ddt_comm_get_unique_id(id) != DDT_OK) return false;
    (void)remove(path.c_str());  // a file left by an earlier job holds a dead id: the launcher starts rank 0 first (or passes a fresh --id-file per job)
    const std::string tmp = path + ".tmp";
    if (!write_file(tmp, id, DDT_COMM_ID_BYTES)) return false;
    return rename(tmp.c_str(), path.c_str()) == 0;
  }
  for (int waited_ms = 0; waited_ms <= timeout_s * 1000; waited_ms += 50) {
    std::vector<unsigned char> buf;
    if (read_file(path, &buf) && buf.size() == DDT_COMM_ID_BYTES) {
      memcpy(id, buf.data(), DDT_COMM_ID_BYTES);
      return true;
    }
    struct timespec ts = {0, 50 * 1000 * 1000};
    nanosleep(&ts, nullptr);
  }
  return false;
}

int die(int rc, ddt_engine* e, const char* what) {
  fprintf(stderr, "ddt_cli: %s: %s%s%s\n", what, ddt_strerror(rc), e ? ": " : "", e ? ddt_last_error(e) : "");
  if (e) ddt_destroy(e);
  return 1;
}

int cmd_gen(const std::map<std::string, std::string>& o) {
  const uint32_t T = (uint32_t)num(o, "trees", 0, true), D = (uint32_t)num(o, "levels", 0, true);
  const uint32_t F = (uint32_t)num(o, "features", 0, true), G = (uint32_t)num(o, "devices", 1);
  const uint64_t N = num(o, "rows", 0, true);
  const int dist = (int)num(o, "dist", 0);
  const std::string prefix = str(o, "prefix");
  ddt_params p;
  memset(&p, 0, sizeof(p));
  p.num_trees = T;
  p.num_levels = D;
  p.num_features = F;
  p.missing_bits = 0x7FC00000u;
  p.weights_lines_per_tree = (uint32_t)((((1ull << (D + 1)) - 1) + 3) / 4);
  p.findex_lines_per_tree = (uint32_t)((((1ull << D) - 1) + 7) / 8);
  const uint32_t per_dev = (T + G - 1) / G;
  p.clusters_per_tuple = per_dev <= 128 ? 1 : per_dev <= 256 ? 2 : per_dev <= 512 ? 4 : 8;
  uint64_t csr[DDT_CSR_COUNT];
  int rc = ddt_csr_encode(&p, N, G, csr);
  if (rc) return die(rc, nullptr, "csr encode");
  std::vector<uint32_t> w((size_t)T * p.weights_lines_per_tree * 4);
  std::vector<uint16_t> f((size_t)T * p.findex_lines_per_tree * 8);
  rc = ddt_synth_model(T, D, F, dist, w.data(), f.data());
  if (rc) return die(rc, nullptr, "synth model");
  const size_t W = (F + 3) / 4 * 4;
  std::vector<uint32_t> x((size_t)N * W);
  rc = ddt_synth_tuples_host(x.data(), 0, N, F, dist, p.missing_bits);
  if (rc) return die(rc, nullptr, "synth tuples");
  FILE* c = fopen((prefix + ".csr").c_str(), "w");
  if (!c) return die(DDT_EINVAL, nullptr, "open csr file");
  fprintf(c, "# soft registers as the host writes them (EngineCSR.sv:190-305); CSR 200 (start) last on the wire\n");
  for (int k = 1; k < DDT_CSR_COUNT; ++k) fprintf(c, "%d %016" PRIx64 "\n", DDT_CSR_FIRST + k, csr[k]);
  fprintf(c, "%d %016" PRIx64 "\n", DDT_CSR_FIRST, csr[0]);
  fclose(c);
  if (!write_file(prefix + ".weights", w.data(), w.size() * 4) || !write_file(prefix + ".findex", f.data(), f.size() * 2) ||
      !write_file(prefix + ".tuples", x.data(), x.size() * 4))
    return die(DDT_EINVAL, nullptr, "write stream files");
  printf("wrote %s.{csr,weights,findex,tuples}: %u trees x depth %u x %u features, %" PRIu64 " tuples, %u device(s)\n",
         prefix.c_str(), T, D, F, N, G);
  return 0;
}

int cmd_score(const std::map<std::string, std::string>& o) {
  uint64_t csr[DDT_CSR_COUNT];
  if (!read_csr(str(o, "csr"), csr)) return die(DDT_EINVAL, nullptr, "read csr file");
  ddt_params p;
  uint64_t n_csr = 0;
  uint32_t devices = 1;
  int rc = ddt_csr_decode(csr, &p, &n_csr, &devices);
  if (rc) return die(rc, nullptr, "csr decode");
  p.cmp_mode = (uint32_t)num(o, "cmp-mode", 0);
  p.sum_mode = (uint32_t)num(o, "sum-mode", 0);
  std::vector<unsigned char> w, f, x;
  if (!read_file(str(o, "weights"), &w) || !read_file(str(o, "findex"), &f) || !read_file(str(o, "tuples"), &x))
    return die(DDT_EINVAL, nullptr, "read stream files");
  const size_t tuple_bytes = (size_t)p.num_features * 4;  // decode returns F = 4 * tuple lines
  const uint64_t n = x.size() / tuple_bytes;              // the tuple stream is authoritative (CSR207 counts whole lines)
  if (x.size() % tuple_bytes) return die(DDT_EINVAL, nullptr, "tuple stream is not a whole number of tuples");
  if ((n + 3) / 4 != (n_csr + 3) / 4)
    fprintf(stderr, "ddt_cli: note: CSR207 announces %" PRIu64 " result lines, the tuple stream holds %" PRIu64 " tuples\n", n_csr / 4, n);
  std::vector<float> scores((size_t)((n + 3) / 4 * 4), 0.0f);  // whole result lines, zero padded
  if (o.count("devices")) {  // the tree-sharded multi-GPU job of the CSR block's mode, all devices driven from this process
    const int G = (int)num(o, "devices", 1);
    const std::string cmb = o.count("combine") ? o.at("combine") : "allreduce";
    if (G < 1 || (cmb != "allreduce" && cmb != "chain")) return die(DDT_EINVAL, nullptr, "--devices / --combine");
    const bool rows_mode = o.count("mode") && o.at("mode") == "rows";  // the ensemble on every device, the tuples partitioned
    const bool hybrid_mode = o.count("mode") && o.at("mode") == "hybrid";
    if (o.count("mode") && !rows_mode && !hybrid_mode && o.at("mode") != "trees") return die(DDT_EINVAL, nullptr, "--mode trees|rows|hybrid");
    const int Gt = hybrid_mode ? (int)num(o, "tree-ranks", 2) : 0;
    if (hybrid_mode && (Gt < 1 || G % Gt)) return die(DDT_EINVAL, nullptr, "--tree-ranks must divide --devices");
    ddt_group* g = nullptr;
    rc = hybrid_mode ? ddt_group_create_hybrid(&g, G, nullptr, Gt) : ddt_group_create(&g, G, nullptr);  // devices 0 .. G-1
    if (rc) return die(rc, nullptr, "ddt_group_create");
    if (o.count("variant"))
      for (int i = 0; i < G; ++i) ddt_set_option(ddt_group_engine(g, i), "variant", (int64_t)num(o, "variant", 0));
    if (rows_mode) {
      rc = ddt_group_load_model_replicated(g, &p, w.data(), w.size() / 16, f.data(), f.size() / 16);
      if (!rc) rc = ddt_group_score_rows(g, x.data(), n, scores.data());
    } else {
      rc = ddt_group_load_model(g, &p, w.data(), w.size() / 16, f.data(), f.size() / 16);
      if (!rc) rc = ddt_group_score(g, x.data(), n, scores.data(), cmb == "chain" ? DDT_COMBINE_CHAIN : DDT_COMBINE_ALLREDUCE);
    }
    if (rc) {
      fprintf(stderr, "ddt_cli: multi-GPU job failed: %s (%s)\n", ddt_strerror(rc), ddt_group_last_error(g));
      ddt_group_destroy(g);
      return 1;
    }
    if (!write_file(str(o, "out"), scores.data(), scores.size() * 4)) return die(DDT_EINVAL, nullptr, "write results");
    ddt_info info;
    ddt_get_info(ddt_group_engine(g, 0), &info);
    warn_fallback(info);
    if (rows_mode)
      printf("scored %" PRIu64 " tuples on %d device(s), %u trees on every device, tuples partitioned (kernel %s), no collective\n", n, G, p.num_trees,
             info.variant_name);
    else if (hybrid_mode)
      printf("scored %" PRIu64 " tuples on %d device(s): %d row group(s) x %d tree shard(s) (device 0: trees [%u, %u), kernel %s), combine %s inside a row group over RCCL\n",
             n, G, G / Gt, Gt, info.tree_begin, info.tree_end, info.variant_name, cmb.c_str());
    else
      printf("scored %" PRIu64 " tuples on %d device(s), %u trees sharded tree-wise (device 0: [%u, %u), kernel %s), combine %s over RCCL\n",
             n, G, p.num_trees, info.tree_begin, info.tree_end, info.variant_name, cmb.c_str());
    ddt_group_destroy(g);
    return 0;
  }
  if (o.count("ranks")) {  // one process per GPU: this process is rank `--rank` of `--ranks`, the id travels through a file
    const int R = (int)num(o, "ranks", 1), r = (int)num(o, "rank", 0, true);
    const std::string cmb = o.count("combine") ? o.at("combine") : "allreduce";
    if (R < 1 || r < 0 || r >= R || (cmb != "allreduce" && cmb != "chain")) return die(DDT_EINVAL, nullptr, "--ranks / --rank / --combine");
    const std::string id_file = str(o, "id-file");
    ddt_engine* e = nullptr;
    rc = ddt_create(&e, (int)num(o, "device", (uint64_t)r));  // default: rank r drives device r
    if (rc) return die(rc, nullptr, "ddt_create");
    if (o.count("variant")) ddt_set_option(e, "variant", (int64_t)num(o, "variant", 0));
    const bool hybrid_mode = o.count("mode") && o.at("mode") == "hybrid";
    if (o.count("mode") && !hybrid_mode && o.at("mode") != "trees") return die(DDT_EINVAL, e, "--ranks: --mode trees|hybrid");
    const int Gt = hybrid_mode ? (int)num(o, "tree-ranks", 2) : R;
    if (Gt < 1 || R % Gt) return die(DDT_EINVAL, e, "--tree-ranks must divide --ranks");
    rc = ddt_load_model_shard(e, &p, w.data(), w.size() / 16, f.data(), f.size() / 16, (uint32_t)(r % Gt), (uint32_t)Gt);
    if (rc) return die(rc, e, "load model shard");
    unsigned char id[DDT_COMM_ID_BYTES];
    if (!exchange_id(id_file, r, (int)num(o, "id-timeout", 120), id)) return die(DDT_ESTATE, e, "communicator id exchange (--id-file)");
    ddt_comm* c = nullptr;
    rc = hybrid_mode ? ddt_comm_create_hybrid(&c, e, r, R, Gt, id) : ddt_comm_create(&c, e, r, R, id);
    if (rc) return die(rc, e, "ddt_comm_create");
    rc = ddt_comm_score(c, x.data(), n, scores.data(), cmb == "chain" ? DDT_COMBINE_CHAIN : DDT_COMBINE_ALLREDUCE);
    if (rc) {
      fprintf(stderr, "ddt_cli: rank %d: sharded job failed: %s (%s)\n", r, ddt_strerror(rc), ddt_comm_last_error(c));
      ddt_comm_destroy(c);
      ddt_destroy(e);
      return 1;
    }
    if (o.count("out") && !write_file(o.at("out"), scores.data(), scores.size() * 4)) return die(DDT_EINVAL, e, "write results");
    ddt_info info;
    ddt_get_info(e, &info);
    warn_fallback(info);
    printf("rank %d of %d%s: scored %" PRIu64 " tuples, trees [%u, %u) of %u on %s, kernel %s, combine %s over RCCL\n", r, R,
           hybrid_mode ? " (hybrid)" : "", n, info.tree_begin, info.tree_end, p.num_trees, info.device_name, info.variant_name, cmb.c_str());
    ddt_comm_destroy(c);
    ddt_destroy(e);
    return 0;
  }
  ddt_engine* e = nullptr;
  rc = ddt_create(&e, (int)num(o, "device", 0));
  if (rc) return die(rc, nullptr, "ddt_create");
  if (o.count("variant")) ddt_set_option(e, "variant", (int64_t)num(o, "variant", 0));
  const uint32_t of = (uint32_t)num(o, "of", 1), shard = (uint32_t)num(o, "shard", 0);
  if (of == 0 || shard >= of) return die(DDT_EINVAL, e, "--shard / --of: need of >= 1 and shard < of");
  rc = ddt_load_model_shard(e, &p, w.data(), w.size() / 16, f.data(), f.size() / 16, shard, of);
  if (rc) return die(rc, e, "load model");
  rc = ddt_score(e, x.data(), n, scores.data());
  if (rc) return die(rc, e, "score");
  if (!write_file(str(o, "out"), scores.data(), scores.size() * 4)) return die(DDT_EINVAL, e, "write results");
  ddt_info info;
  ddt_stats st;
  ddt_get_info(e, &info);
  warn_fallback(info);
  ddt_get_stats(e, &st);
  printf("scored %" PRIu64 " tuples with trees [%u, %u) of %u on %s, kernel %s, %.3f ms (%.2f Mtuples/s incl. PCIe), %" PRIu64
         " result lines\n",
         n, info.tree_begin, info.tree_end, p.num_trees, info.device_name, info.variant_name, st.exec_ms,
         st.exec_ms > 0 ? (double)n / st.exec_ms / 1e3 : 0.0, st.result_lines_out);
  ddt_destroy(e);
  return 0;
}

int cmd_gen_sparse(const std::map<std::string, std::string>& o) {
  const uint32_t T = (uint32_t)num(o, "trees", 0, true), D = (uint32_t)num(o, "max-depth", 0, true);
  const uint32_t F = (uint32_t)num(o, "features", 0, true);
  const uint32_t full = (uint32_t)num(o, "full-levels", D < 10 ? D / 2 : 10), pm = (uint32_t)num(o, "permille", 700);
  const uint64_t N = num(o, "rows", 0, true);
  const int dist = (int)num(o, "dist", 0);
  const std::string prefix = str(o, "prefix");
  std::vector<uint64_t> first((size_t)T + 1);
  const int64_t n_lines = ddt_synth_sparse_model(T, D, F, full, pm, dist, nullptr, 0, first.data());
  if (n_lines < 0) return die((int)n_lines, nullptr, "synth sparse model");
  std::vector<uint32_t> nodes((size_t)n_lines * 4);
  ddt_synth_sparse_model(T, D, F, full, pm, dist, nodes.data(), (size_t)n_lines, first.data());
  const size_t W = (F + 3) / 4 * 4;
  std::vector<uint32_t> x((size_t)N * W);
  int rc = ddt_synth_tuples_host(x.data(), 0, N, F, dist, 0x7FC00000u);
  if (rc) return die(rc, nullptr, "synth tuples");
  if (!write_file(prefix + ".nodes", nodes.data(), nodes.size() * 4) || !write_file(prefix + ".first", first.data(), first.size() * 8) ||
      !write_file(prefix + ".tuples", x.data(), x.size() * 4))
    return die(DDT_EINVAL, nullptr, "write stream files");
  printf("wrote %s.{nodes,first,tuples}: %u sparse trees (depth <= %u, %" PRId64 " internal nodes) x %u features, %" PRIu64 " tuples\n",
         prefix.c_str(), T, D, n_lines, F, N);
  return 0;
}

int cmd_score_sparse(const std::map<std::string, std::string>& o) {
  std::vector<unsigned char> nodes, first, x;
  if (!read_file(str(o, "nodes"), &nodes) || !read_file(str(o, "first"), &first) || !read_file(str(o, "tuples"), &x))
    return die(DDT_EINVAL, nullptr, "read stream files");
  if (first.size() < 16 || first.size() % 8 || nodes.size() % 16) return die(DDT_EINVAL, nullptr, "malformed .first / .nodes file");
  ddt_params p;
  memset(&p, 0, sizeof(p));
  p.num_trees = (uint32_t)(first.size() / 8 - 1);
  p.num_levels = (uint32_t)num(o, "max-depth", 0, true);
  p.num_features = (uint32_t)num(o, "features", 0, true);
  p.missing_bits = (uint32_t)num(o, "missing", 0x7FC00000u);
  p.cmp_mode = (uint32_t)num(o, "cmp-mode", 0);
  p.sum_mode = (uint32_t)num(o, "sum-mode", 0);
  const uint32_t of = (uint32_t)num(o, "of", 1), shard = (uint32_t)num(o, "shard", 0);
  if (of == 0 || shard >= of) return die(DDT_EINVAL, nullptr, "--shard / --of: need of >= 1 and shard < of");
  const uint32_t per_dev = (p.num_trees + of - 1) / of;
  p.clusters_per_tuple = (uint32_t)num(o, "clusters", per_dev <= 128 ? 1 : per_dev <= 256 ? 2 : per_dev <= 512 ? 4 : 8);
  const size_t tuple_bytes = (size_t)(p.num_features + 3) / 4 * 16;
  if (x.size() % tuple_bytes) return die(DDT_EINVAL, nullptr, "tuple stream is not a whole number of tuples");
  const uint64_t n = x.size() / tuple_bytes;
  ddt_engine* e = nullptr;
  int rc = ddt_create(&e, (int)num(o, "device", 0));
  if (rc) return die(rc, nullptr, "ddt_create");
  rc = ddt_load_model_sparse(e, &p, nodes.data(), nodes.size() / 16, reinterpret_cast<const uint64_t*>(first.data()), shard, of);
  if (rc) return die(rc, e, "load sparse model");
  std::vector<float> scores((size_t)((n + 3) / 4 * 4), 0.0f);  // whole result lines, zero padded
  rc = ddt_score(e, x.data(), n, scores.data());
  if (rc) return die(rc, e, "score");
  if (!write_file(str(o, "out"), scores.data(), scores.size() * 4)) return die(DDT_EINVAL, e, "write results");
  ddt_info info;
  ddt_stats st;
  ddt_get_info(e, &info);
  warn_fallback(info);
  ddt_get_stats(e, &st);
  printf("scored %" PRIu64 " tuples with sparse trees [%u, %u) of %u on %s, kernel %s, %.3f ms (%.2f Mtuples/s incl. PCIe)\n", n,
         info.tree_begin, info.tree_end, p.num_trees, info.device_name, info.variant_name, st.exec_ms,
         st.exec_ms > 0 ? (double)n / st.exec_ms / 1e3 : 0.0);
  ddt_destroy(e);
  return 0;
}

int cmd_info() {
  ddt_engine* e = nullptr;
  const int rc = ddt_create(&e, 0);
  printf("libddt ABI %d, %d kernel variants\n", DDT_ABI_VERSION, ddt_num_variants());
  for (int v = 0; v < ddt_num_variants(); ++v) {
    char name[64];
    ddt_variant_name(v, name, sizeof(name));
    printf("  variant %2d  %s\n", v, name);
  }
  if (rc) {
    printf("device: none usable (%s)\n", ddt_strerror(rc));
    return 0;
  }
  ddt_info info;
  ddt_get_info(e, &info);
  printf("device 0: %s\n", info.device_name);
  ddt_destroy(e);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: ddt_cli gen|score|gen-sparse|score-sparse|info [--option value ...]   (see the header of ddt_cli.cpp)\n");
    return 2;
  }
  const std::string cmd = argv[1];
  const auto o = parse(argc, argv, 2);
  if (cmd == "gen") return cmd_gen(o);
  if (cmd == "score") return cmd_score(o);
  if (cmd == "gen-sparse") return cmd_gen_sparse(o);
  if (cmd == "score-sparse") return cmd_score_sparse(o);
  if (cmd == "info") return cmd_info();
  fprintf(stderr, "unknown command %s\n", cmd.c_str());
  return 2;
}
