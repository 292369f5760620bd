// Does a hipMemsetAsync captured into a HIP graph zero its range on EVERY replay, in order with the kernels around it?  (round 6: a captured
// ddt_score_device replayed stale ranks -- the ticket counters of the LDS-resident rank pre-pass, zeroed by a hipMemsetAsync in front of the kernel, were
// not zero on replay.)  Kernel: every wave takes tickets from an 8-byte counter until `units` is reached and adds 1 to out[ticket]; a second memset form
// (a kernel that zeroes) is the control.  Prints per replay: counter before / after, sum of out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void take(unsigned long long* counter, unsigned long long units, unsigned* out) {
  for (;;) {
    unsigned long long u = 0;
    if ((threadIdx.x & 63) == 0) u = atomicAdd(counter, 1ull);
    u = __shfl(u, 0);
    if (u >= units) break;
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[u], 1u);
  }
}
__global__ void zero(unsigned* p, unsigned n) { for (unsigned i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0; }

int run(size_t memset_bytes, bool kernel_zero, size_t counter_off_words) {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  unsigned* buf;  // [flags ... | counter]
  const unsigned units = 1000;
  unsigned* out;
  CK(hipMalloc(&buf, 1 << 16));
  CK(hipMalloc(&out, units * 4));
  CK(hipMemset(buf, 0xFF, 1 << 16));
  unsigned long long* counter = reinterpret_cast<unsigned long long*>(buf + counter_off_words);
  auto enqueue = [&]() -> hipError_t {
    hipError_t e = hipSuccess;
    if (kernel_zero) {
      hipLaunchKernelGGL(zero, dim3(1), dim3(256), 0, s, buf, (unsigned)(memset_bytes / 4));
      e = hipGetLastError();
    } else {
      e = hipMemsetAsync(buf, 0, memset_bytes, s);
    }
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(take, dim3(64), dim3(256), 0, s, counter, (unsigned long long)units, out);
    return hipGetLastError();
  };
  CK(hipMemsetAsync(out, 0, units * 4, s));
  CK(enqueue());
  CK(hipStreamSynchronize(s));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  CK(enqueue());
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  size_t nn = 0;
  CK(hipGraphGetNodes(g, nullptr, &nn));
  printf("memset %zu B, %s, counter at word %zu: graph of %zu nodes\n", memset_bytes, kernel_zero ? "zeroing KERNEL" : "hipMemsetAsync", counter_off_words, nn);
  std::vector<unsigned> h(units);
  for (int r = 0; r < 3; ++r) {
    unsigned long long before = 0, after = 0;
    CK(hipMemcpy(&before, counter, 8, hipMemcpyDeviceToHost));
    CK(hipMemsetAsync(out, 0, units * 4, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(&after, counter, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h.data(), out, units * 4, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    for (unsigned v : h) sum += v;
    printf("  replay %d: counter before %llu after %llu, units done %llu of %u\n", r, before, after, sum, units);
  }
  return 0;
}

int main() {
  run(528, false, 2);      // two tiles: flags[2] | counters (the small batch of the test)
  run(1304, false, 196);   // 196 tiles
  run(528, true, 2);
  run(1304, true, 196);
  run(4096, false, 196);
  return 0;
}
