// A captured ddt_score_device, replayed on new tuples in the same buffer, WITHOUT torch: plain hipStreamBeginCapture around the C-ABI call.
// Prints per replay whether the scores equal those of a direct call on the same tuples (bitwise).  Modes (bits): 1 thread-local capture, 2 replay on the null
// stream with the tuples copied from pageable host memory, 4 warm-up call on the null stream, 8 hipGraphInstantiateWithFlags(AutoFreeOnLaunch).
// Build (from tools/ubench): hipcc --offload-arch=gfx950 -O2 -I../../include -o graph_capi graph_capi.cpp -L../../distributed-decisiontrees_amd/lib -lddt \
//   -Wl,-rpath,'$ORIGIN/../../distributed-decisiontrees_amd/lib';  graph_memset.hip: hipcc --offload-arch=gfx950 -O2 -o graph_memset graph_memset.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "ddt.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define DK(x) do { int r_ = (x); if (r_) { printf("%s -> %d (%s)\n", #x, r_, ddt_last_error(e)); return 1; } } while (0)

int run(uint32_t T, uint32_t D, uint32_t F, size_t n, const char* opt, long long val, int mode) {
  ddt_engine* e = nullptr;
  if (ddt_create(&e, 0)) return 1;
  if (opt) DK(ddt_set_option(e, opt, val));
  ddt_params p;
  memset(&p, 0, sizeof p);
  p.num_trees = T, p.num_levels = D, p.num_features = F, p.missing_bits = 0x7FC00000u;
  p.weights_lines_per_tree = ((2u << D) - 1u + 3u) / 4u, p.findex_lines_per_tree = ((1u << D) - 1u + 7u) / 8u;
  p.clusters_per_tuple = 8;
  std::vector<uint32_t> w((size_t)T * p.weights_lines_per_tree * 4);
  std::vector<uint16_t> f((size_t)T * p.findex_lines_per_tree * 8);
  DK(ddt_synth_model(T, D, F, 0, w.data(), f.data()));
  DK(ddt_load_model(e, &p, w.data(), w.size() / 4, f.data(), f.size() / 8));
  const size_t W = (F + 3) / 4 * 4;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  void* d;
  float *out, *ref;
  CK(hipMalloc(&d, n * W * 4));
  CK(hipMalloc(&out, n * 4));
  CK(hipMalloc(&ref, n * 4));
  DK(ddt_synth_tuples_device(e, d, 0, n, F, 0, p.missing_bits, s));
  DK(ddt_score_device(e, d, n, out, (mode & 4) ? nullptr : s));  // sizes the workspaces (mode bit 2: on the null stream, as the torch test's warm-up call)
  CK(hipDeviceSynchronize());
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, (mode & 1) ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeGlobal));
  DK(ddt_score_device(e, d, n, out, s));
  CK(hipStreamEndCapture(s, &g));
  if (mode & 8) CK(hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagAutoFreeOnLaunch));  // (what PyTorch's CUDAGraph does)
  else CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  size_t nn = 0;
  CK(hipGraphGetNodes(g, nullptr, &nn));
  ddt_info info;
  DK(ddt_get_info(e, &info));
  printf("%u x d%u x %u f, %zu rows, %s=%lld, kernel %s, pre-pass groups %u: graph of %zu nodes;", T, D, F, n, opt ? opt : "-", val, info.variant_name, info.prepass_groups, nn);
  std::vector<uint32_t> a(n), b(n);
  for (int r = 1; r <= 3; ++r) {
    hipStream_t rs = (mode & 2) ? nullptr : s;  // mode bit 1: replay on the null stream, tuples by a pageable hipMemcpy (what the torch test does)
    if (mode & 2) {
      std::vector<uint32_t> h(n * W);
      DK(ddt_synth_tuples_host(h.data(), 1000003ull * r, n, F, 0, p.missing_bits));
      CK(hipMemcpy(d, h.data(), n * W * 4, hipMemcpyHostToDevice));
      CK(hipDeviceSynchronize());
    } else {
      DK(ddt_synth_tuples_device(e, d, 1000003ull * r, n, F, 0, p.missing_bits, s));
    }
    CK(hipMemsetAsync(out, 0, n * 4, rs));
    CK(hipGraphLaunch(ge, rs));
    CK(hipStreamSynchronize(rs));
    DK(ddt_score_device(e, d, n, ref, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(a.data(), out, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), ref, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += a[i] != b[i];
    printf(" replay %d: %zu rows differ;", r, bad);
  }
  printf("\n");
  ddt_destroy(e);
  return 0;
}

int main() {
  run(300, 8, 32, 2000, nullptr, 0, 0);
  run(300, 8, 32, 2000, "q16_cluster_split", 0, 0);
  run(300, 8, 32, 2000, "q16_grouped_prepass", 0, 0);
  run(300, 8, 32, 200000, nullptr, 0, 0);
  run(300, 8, 32, 200000, nullptr, 0, 1);
  run(125, 8, 32, 200000, nullptr, 0, 0);
  run(30, 6, 16, 5000, nullptr, 0, 0);
  run(300, 8, 32, 2000, nullptr, 0, 2);
  run(300, 8, 32, 200000, nullptr, 0, 2);
  run(300, 8, 32, 200000, "q16_grouped_prepass", 0, 2);
  run(300, 8, 32, 2000, nullptr, 0, 6);
  run(300, 8, 32, 200000, nullptr, 0, 6);
  run(300, 8, 32, 2000, nullptr, 0, 14);
  run(300, 8, 32, 200000, nullptr, 0, 14);
  return 0;
}
