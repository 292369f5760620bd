// ddt_comm.cpp -- multi-GPU jobs behind the C-ABI: RCCL (librccl, the ROCm NCCL) over xGMI.
//
// Replaces the reference's inter-FPGA networks:
//   tuple broadcast            rtl/DTEngine/InputDistributor.sv:199-204 (ring re-broadcast of tuple lines)
//   partial-result aggregation rtl/DTEngine/ResultsCombiner.sv:292-311 (four fp32 adders: local line + upstream line),
//                              :359-369,426-430 (forwarding along host -> dev1 -> ... and back to the host's PCIe)
//   per-device tree shards     rtl/DTEngine/PCIeReceiver.sv:241-264 (CSR 203), tuple batches :289-312 (row mode)
// MI355X mapping: one communicator rank per GPU; the tree-sharded job runs a chunk pipeline on two HIP streams -- the
// caller's stream scores chunk k+1 while the comm's own stream combines chunk k -- with either one ncclAllReduce per
// chunk (the collective BASELINE.json names) or the deterministic chain: grouped ncclSend/ncclRecv all-to-all of 1/G
// slices, fixed-order add on the owner, ncclAllGather.  xGMI is point-to-point (fully connected mesh), so the
// all-to-all form puts one slice on every link at once.  The other two jobs are written for the mesh the same way: the row-sharded
// replicas hand every finished step of scores to all peers with grouped send / recv while the next step is scored, and host tuples
// cross PCIe once (1/n per rank) before the ranks hand their rows to each other (tuples_to_device).
// The two modes composed (the "hybrid", ddt_comm_create_hybrid): the n ranks form n / Gt row groups of Gt consecutive ranks; a row
// group is a tree-sharded job of its own on ITS slice of the rows (shard r % Gt of Gt; all-reduce / chain inside the group, on a
// communicator split off the world communicator), and the finished pieces are handed to the other row groups over the world
// communicator while the next piece is scored.  What it buys at 8 GPUs: the replicated work of the tree-sharded mode (every rank
// ranks ALL tuples against its thresholds; every rank reads all tuples) shrinks by the number of row groups, the all-reduce spans Gt
// ranks instead of n (DTInference.sv:28-37 has both modes; PCIeReceiver.sv:241-264,289-312 the two splits; ResultsCombiner.sv:292-311
// adds, :371-391 interleaves).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "ddt_engine_priv.h"

using namespace ddt;

struct ddt_comm {
  ddt_engine* e = nullptr;
  int device = 0;  // e->device at creation: the teardown must not read an engine that may already be gone
  int rank = 0, n = 1;               // rank inside / size of `comm`: the communicator the partial scores are combined over
  ncclComm_t comm = nullptr;
  // hybrid jobs (ddt_comm_create_hybrid): `comm` is the ROW GROUP's communicator (rank = tree shard index, n = Gt), `world` spans all
  // ranks; a plain communicator has world == nullptr, row_groups == 1
  ncclComm_t world = nullptr;
  int wrank = 0, wn = 1;
  int row_groups = 1, row_group = 0;
  bool dead = false;                 // ddt_comm_abort: every later call is refused
  hipStream_t cs = nullptr;                       // the comm's own stream: collectives + chain adds
  hipEvent_t ev_scored[2] = {nullptr, nullptr};   // caller's stream: chunk scored into slot b
  hipEvent_t ev_free[2] = {nullptr, nullptr};     // comm stream: slot b consumed
  hipEvent_t ev_done = nullptr;
  size_t chunk_rows = 12'500'000;
  int taper_tail = -1;               // option "taper_tail": 1 = split the last chunk (1/2, 1/4, 1/4), 0 = never, -1 = when n > 1
  size_t taper_min_rows = 1u << 20;  // option "taper_min_rows": pieces are not made smaller than this
  // chain / classify workspaces, two slots (grow-only): part = this rank's partial values (G*seg floats, zero padded),
  // recv = every rank's slice of my segment [G][seg], full = combined values of all segments
  float* part[2] = {nullptr, nullptr};
  float* recv[2] = {nullptr, nullptr};
  float* full[2] = {nullptr, nullptr};
  size_t cap = 0;  // floats per buffer
  bool slot_used[2] = {false, false};
  // host-buffer form (ddt_comm_score): this rank's staging buffers and stream, grow-only
  hipStream_t hs = nullptr;
  void* h_tuples = nullptr;
  float* h_scores = nullptr;
  size_t h_rows = 0, h_words = 0;
  size_t host_rows = 8u << 20;  // option "host_rows": rows per super-chunk held on the device at once
  int tuple_broadcast = -1;     // option "tuple_broadcast": host tuples cross PCIe once (1/n per rank) and are handed to the peers over
                                // xGMI; 0 = every rank copies all of them from the host; -1 = on when n > 1
  char err[256] = {0};
};

namespace {

int cfail(ddt_comm* c, int code, const char* fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define CHIP(c, call)                                                                             \
  do {                                                                                            \
    hipError_t _r = (call);                                                                       \
    if (_r != hipSuccess) return cfail((c), DDT_EHIP, "%s -> %s", #call, hipGetErrorString(_r));  \
  } while (0)
#define CNCCL(c, call)                                                                            \
  do {                                                                                            \
    ncclResult_t _r = (call);                                                                     \
    if (_r != ncclSuccess) return cfail((c), DDT_EHIP, "%s -> %s", #call, ncclGetErrorString(_r)); \
  } while (0)

// closes the NCCL group when a CNCCL / CHIP in the middle of it returns early: an open group would swallow every later call on this thread
struct GroupGuard {
  bool open = false;
  ncclResult_t start() {
    const ncclResult_t r = ncclGroupStart();
    open = r == ncclSuccess;
    return r;
  }
  ncclResult_t end() {
    open = false;
    return ncclGroupEnd();
  }
  ~GroupGuard() {
    if (open) (void)ncclGroupEnd();
  }
};

// rows of a host-buffer super-chunk: the option, but never more than 2 GiB of tuples on the device at once (8 Mi rows of 2048
// features would ask for 64 GiB); whole 1024-tuple tiles
size_t host_rows_cap(size_t words) { return std::max<size_t>(((size_t)2u << 30) / (words * 4u) / 1024u * 1024u, 1024u); }

int comm_init_common(ddt_comm* c) {
  CHIP(c, hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking));
  for (int b = 0; b < 2; ++b) {
    CHIP(c, hipEventCreateWithFlags(&c->ev_scored[b], hipEventDisableTiming));
    CHIP(c, hipEventCreateWithFlags(&c->ev_free[b], hipEventDisableTiming));
  }
  CHIP(c, hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  return DDT_OK;
}

// workspaces for `floats` values per chunk (rounded up to whole segments by the callers)
int comm_reserve(ddt_comm* c, size_t floats) {
  if (floats <= c->cap) return DDT_OK;
  CHIP(c, hipDeviceSynchronize());
  for (int b = 0; b < 2; ++b)
    for (float** p : {&c->part[b], &c->recv[b], &c->full[b]}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
  c->cap = 0;
  for (int b = 0; b < 2; ++b)
    for (float** p : {&c->part[b], &c->recv[b], &c->full[b]}) CHIP(c, hipMalloc(reinterpret_cast<void**>(p), floats * sizeof(float)));
  c->cap = floats;
  c->slot_used[0] = c->slot_used[1] = false;
  return DDT_OK;
}

// Combine `count` partial values sitting in part[b][0, count) (tail up to G*seg already zero) into full[b][0, count),
// on the comm stream.  seg = ceil(count / G).
int chain_combine(ddt_comm* c, int b, size_t count) {
  const size_t G = (size_t)c->n, seg = (count + G - 1) / G;
  // all-to-all: my slice r goes to rank r; I receive every rank's slice of MY segment
  GroupGuard grp;
  CNCCL(c, grp.start());
  for (int r = 0; r < c->n; ++r) {
    CNCCL(c, ncclSend(c->part[b] + (size_t)r * seg, seg, ncclFloat, r, c->comm, c->cs));
    CNCCL(c, ncclRecv(c->recv[b] + (size_t)r * seg, seg, ncclFloat, r, c->comm, c->cs));
  }
  CNCCL(c, grp.end());
  // p0 + p1 + ... in rank order: the reference's hop order (ResultsCombiner.sv:292-311: local + upstream)
  hipError_t r = launch_chain_sum(c->recv[b], (uint32_t)c->n, seg, c->full[b] + (size_t)c->rank * seg, c->e && c->e->p.sum_mode == 2, c->cs);
  if (r != hipSuccess) return cfail(c, DDT_EHIP, "chain_sum -> %s", hipGetErrorString(r));
  CNCCL(c, ncclAllGather(c->full[b] + (size_t)c->rank * seg, c->full[b], seg, ncclFloat, c->comm, c->cs));
  return DDT_OK;
}

int check_call(ddt_comm* c, const void* d_tuples, const void* d_out, size_t n) {
  if (!c) return DDT_EINVAL;
  if (c->dead) return cfail(c, DDT_ESTATE, "communicator was aborted");
  if (!c->e || !c->e->loaded) return cfail(c, DDT_ESTATE, "no model loaded on the engine of this communicator");
  if (n && (!d_tuples || !d_out)) return cfail(c, DDT_EINVAL, "NULL device buffer");
  return DDT_OK;
}

// Chunk lengths of a sharded call: `rows` each; with `taper` the final stretch (<= rows) is cut into halves down to
// max(rows / 4, min_rows), whole 1024-tuple tiles, so that the collective left exposed behind the last scoring launch is about a
// quarter of a chunk.  Every rank computes the same list from the same arguments.
std::vector<size_t> chunk_schedule(size_t n, size_t rows, bool taper, size_t min_rows) {
  std::vector<size_t> out;
  rows = rows ? rows : 1;
  const size_t floor_rows = std::max(rows / 4, std::max<size_t>(min_rows, 1));
  for (size_t left = n; left;) {
    size_t m = std::min(rows, left);
    if (taper && left <= rows && left > floor_rows) {
      size_t half = left / 2;
      if (half >= 1024) half = (half + 1023) / 1024 * 1024;  // whole tiles of the scoring kernels
      if (half && half < left) m = half;
    }
    out.push_back(m);
    left -= m;
  }
  return out;
}

// rows [lo, hi) of row group rg of Gr: equal slices in whole 1024-tuple tiles (whole result lines: ResultsCombiner.sv:371-391 hands
// out whole lines too), the last group takes what is left, groups past the end are empty
void hybrid_window(size_t n, size_t Gr, size_t rg, size_t* lo, size_t* hi) {
  size_t per = (n + Gr - 1) / Gr;
  per = (per + 1023) / 1024 * 1024;
  *lo = std::min(n, rg * per);
  *hi = std::min(n, (rg + 1) * per);
}

}  // namespace

extern "C" {

int ddt_hybrid_rows(size_t n_tuples, int row_groups, int row_group, size_t* lo, size_t* hi) {
  if (row_groups < 1 || row_group < 0 || row_group >= row_groups || !lo || !hi) return DDT_EINVAL;
  hybrid_window(n_tuples, (size_t)row_groups, (size_t)row_group, lo, hi);
  return DDT_OK;
}

int ddt_comm_get_unique_id(void* id_out) {
  if (!id_out) return DDT_EINVAL;
  static_assert(sizeof(ncclUniqueId) == DDT_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return DDT_EHIP;
  memcpy(id_out, &id, sizeof(id));
  return DDT_OK;
}

// tree_ranks == 0: a plain communicator over the n ranks.  tree_ranks = Gt >= 1 (a divisor of n_ranks): the hybrid layout -- the world
// communicator from the unique id, the row group's communicator split off it (colour = row group, key = shard index: all ranks make
// the same ncclCommSplit call, RCCL's own rendezvous; no second id has to travel)
static int comm_create(ddt_comm** out, ddt_engine* e, int rank, int n_ranks, int tree_ranks, const void* unique_id) {
  if (!out) return DDT_EINVAL;
  *out = nullptr;
  if (!e || !unique_id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return DDT_EINVAL;
  if (tree_ranks < 0 || (tree_ranks > 0 && n_ranks % tree_ranks != 0)) return fail(e, DDT_EINVAL, "tree_ranks %d does not divide %d ranks", tree_ranks, n_ranks);
  std::unique_ptr<ddt_comm> c(new (std::nothrow) ddt_comm());
  if (!c) return DDT_ENOMEM;
  c->e = e;
  c->device = e->device;
  c->rank = c->wrank = rank;
  c->n = c->wn = n_ranks;
  DeviceGuard dg(e->device);
  if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, id, rank);
  if (r != ncclSuccess) return fail(e, DDT_EHIP, "ncclCommInitRank(rank %d of %d) -> %s", rank, n_ranks, ncclGetErrorString(r));
  if (tree_ranks > 0) {
    c->world = c->comm;
    c->comm = nullptr;
    c->row_groups = n_ranks / tree_ranks;
    c->row_group = rank / tree_ranks;
    c->rank = rank % tree_ranks;
    c->n = tree_ranks;
    r = ncclCommSplit(c->world, c->row_group, c->rank, &c->comm, nullptr);
    if (r != ncclSuccess || !c->comm) {
      (void)ncclCommDestroy(c->world);
      return fail(e, DDT_EHIP, "ncclCommSplit(rank %d: row group %d, shard %d of %d) -> %s", rank, c->row_group, c->rank, tree_ranks, ncclGetErrorString(r));
    }
  }
  int rc = comm_init_common(c.get());
  if (rc) {
    fail(e, rc, "%s", c->err);
    (void)ncclCommDestroy(c->comm);
    if (c->world) (void)ncclCommDestroy(c->world);
    return rc;
  }
  if (n_ranks > 1) engine_enter_collective_job(e);  // collectives will share the device's CUs with the scoring kernels
  *out = c.release();
  return DDT_OK;
}

int ddt_comm_create(ddt_comm** out, ddt_engine* e, int rank, int n_ranks, const void* unique_id) { return comm_create(out, e, rank, n_ranks, 0, unique_id); }

int ddt_comm_create_hybrid(ddt_comm** out, ddt_engine* e, int rank, int n_ranks, int tree_ranks, const void* unique_id) {
  if (tree_ranks < 1) return e ? fail(e, DDT_EINVAL, "tree_ranks must be >= 1") : DDT_EINVAL;
  return comm_create(out, e, rank, n_ranks, tree_ranks, unique_id);
}

int ddt_comm_layout(const ddt_comm* c, ddt_comm_layout_t* out) {
  if (!c || !out) return DDT_EINVAL;
  out->rank = c->wrank;
  out->n_ranks = c->wn;
  out->tree_ranks = c->n;
  out->tree_rank = c->rank;
  out->row_groups = c->row_groups;
  out->row_group = c->row_group;
  return DDT_OK;
}

// A peer failed before (or in) its collective: the survivors' streams would wait for ever.  ncclCommAbort frees them; the
// communicator is dead afterwards (every call is refused), ddt_comm_destroy still releases it.
int ddt_comm_abort(ddt_comm* c) {
  if (!c) return DDT_EINVAL;
  if (c->dead) return DDT_OK;
  DeviceGuard dg(c->device);
  c->dead = true;
  ncclResult_t r1 = c->comm ? ncclCommAbort(c->comm) : ncclSuccess;
  ncclResult_t r2 = c->world ? ncclCommAbort(c->world) : ncclSuccess;
  c->comm = c->world = nullptr;
  if (r1 != ncclSuccess || r2 != ncclSuccess) return cfail(c, DDT_EHIP, "ncclCommAbort -> %s", ncclGetErrorString(r1 != ncclSuccess ? r1 : r2));
  return DDT_OK;
}

void ddt_comm_destroy(ddt_comm* c) {
  if (!c) return;
  DeviceGuard dg(c->device);  // not c->e->device: a caller may have destroyed the engine first
  if (c->cs) (void)hipStreamSynchronize(c->cs);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->world) (void)ncclCommDestroy(c->world);
  for (int b = 0; b < 2; ++b) {
    for (float** p : {&c->part[b], &c->recv[b], &c->full[b]})
      if (*p) (void)hipFree(*p);
    if (c->ev_scored[b]) (void)hipEventDestroy(c->ev_scored[b]);
    if (c->ev_free[b]) (void)hipEventDestroy(c->ev_free[b]);
  }
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->cs) (void)hipStreamDestroy(c->cs);
  if (c->hs) {
    (void)hipStreamSynchronize(c->hs);
    (void)hipStreamDestroy(c->hs);
  }
  if (c->h_tuples) (void)hipFree(c->h_tuples);
  if (c->h_scores) (void)hipFree(c->h_scores);
  delete c;
}

const char* ddt_comm_last_error(const ddt_comm* c) {
  if (!c) return "";
  return c->err;  // (engine failures are copied into c->err where they happen)
}

int ddt_comm_set_option(ddt_comm* c, const char* key, int64_t value) {
  if (!c || !key) return DDT_EINVAL;
  if (!strcmp(key, "chunk_rows")) {
    if (value < 1) return cfail(c, DDT_EINVAL, "chunk_rows must be >= 1");
    c->chunk_rows = (size_t)value;
    return DDT_OK;
  }
  if (!strcmp(key, "comm_stream_priority")) {  // 1 = the comm stream gets the device's highest stream priority (collective blocks are
    // dispatched ahead of the scoring launch's queued blocks as CUs free up), 0 = default priority
    if (value != 0 && value != 1) return cfail(c, DDT_EINVAL, "comm_stream_priority must be 0 or 1");
    DeviceGuard dg(c->device);
    hipStream_t ns = nullptr;
    if (value) {
      int least = 0, greatest = 0;
      CHIP(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
      CHIP(c, hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, greatest));
    } else {
      CHIP(c, hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
    }
    if (c->cs) {
      CHIP(c, hipStreamSynchronize(c->cs));
      (void)hipStreamDestroy(c->cs);
    }
    c->cs = ns;
    return DDT_OK;
  }
  if (!strcmp(key, "tuple_broadcast")) {
    if (value < -1 || value > 1) return cfail(c, DDT_EINVAL, "tuple_broadcast must be -1 (automatic), 0 or 1");
    c->tuple_broadcast = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "host_rows")) {
    if (value < 1) return cfail(c, DDT_EINVAL, "host_rows must be >= 1");
    c->host_rows = (size_t)value;
    return DDT_OK;
  }
  if (!strcmp(key, "taper_tail")) {
    if (value < -1 || value > 1) return cfail(c, DDT_EINVAL, "taper_tail must be -1 (automatic), 0 or 1");
    c->taper_tail = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "taper_min_rows")) {
    if (value < 1) return cfail(c, DDT_EINVAL, "taper_min_rows must be >= 1");
    c->taper_min_rows = (size_t)value;
    return DDT_OK;
  }
  return cfail(c, DDT_EINVAL, "unknown option '%s'", key);
}

int64_t ddt_comm_chunk_schedule(size_t n, size_t chunk_rows, int taper, size_t taper_min_rows, size_t* lens_out, size_t cap) {
  if (chunk_rows == 0 || n / chunk_rows > (1u << 20)) return DDT_EINVAL;
  const std::vector<size_t> v = chunk_schedule(n, chunk_rows, taper != 0, taper_min_rows);
  if (lens_out) {
    if (cap < v.size()) return DDT_EINVAL;
    std::copy(v.begin(), v.end(), lens_out);
  }
  return (int64_t)v.size();
}

// Shared chunk pipeline.  K = values per row (1 = scores, num_classes = class sums); `dst` = [K][n_total] result.
// The job covers rows [lo, hi) of a batch of n_total rows (a plain communicator: all of them): the tuples are read at row lo + ...,
// the results land at dst[k * n_total + row].  gather (hybrid communicators): every finished piece is handed to the ranks that hold
// the same tree shard in the other row groups (world rank = row group * Gt + shard: one message per peer and class, straight into
// place) and theirs are received -- on the comm stream, behind the piece's combine, while the next piece is scored; every rank
// ends up with all n_total rows.
static int sharded_pipeline(ddt_comm* c, const void* d_tuples, size_t n_total, size_t lo_row, size_t hi_row, float* dst, uint32_t K, int combine,
                            bool gather, hipStream_t s) {
  ddt_engine* e = c->e;
  const size_t W = tuple_words(e->p), G = (size_t)c->n, n = hi_row - lo_row;
  const bool chain = combine == DDT_COMBINE_CHAIN;
  const bool staged = chain || K > 1;  // all-reduce of plain scores runs in place in the caller's buffer
  const size_t rows = std::min(c->chunk_rows, n);
  c->err[0] = 0;
  if (staged && n) {
    const size_t seg = (rows * K + G - 1) / G;
    int rc = comm_reserve(c, seg * G);
    if (rc) return rc;
  }
  const uint32_t* tup = reinterpret_cast<const uint32_t*>(d_tuples);
  const bool taper = c->taper_tail < 0 ? c->n > 1 : c->taper_tail != 0;
  // a schedule of more than 2^20 pieces is a mis-set chunk_rows, not a job (and its vector could throw across the C ABI)
  if (c->chunk_rows && n_total / c->chunk_rows > (1u << 20)) return cfail(c, DDT_EINVAL, "chunk_rows %zu cuts %zu rows into more than 2^20 chunks", c->chunk_rows, n_total);
  const std::vector<size_t> sched = chunk_schedule(n, rows, taper, c->taper_min_rows);  // every piece <= rows: fits the workspaces
  // gather: the pieces of every row group (each derives every group's list from the same arguments)
  gather = gather && c->row_groups > 1;
  const size_t Gr = (size_t)c->row_groups;
  std::vector<std::vector<size_t>> peer_sched;
  std::vector<size_t> peer_lo;
  size_t steps = sched.size();
  if (gather) {
    peer_sched.resize(Gr);
    peer_lo.resize(Gr);
    for (size_t r = 0; r < Gr; ++r) {
      size_t l, h;
      hybrid_window(n_total, Gr, r, &l, &h);
      peer_lo[r] = l;
      peer_sched[r] = chunk_schedule(h - l, std::min(c->chunk_rows, h - l), taper, c->taper_min_rows);
      steps = std::max(steps, peer_sched[r].size());
    }
  }
  std::vector<size_t> peer_off(gather ? Gr : 0, 0);  // rows of each group's earlier pieces
  size_t lo = lo_row;
  for (size_t k = 0; k < steps; ++k) {
    const int b = (int)(k & 1);
    if (k < sched.size()) {
      const size_t m = sched[k], count = m * K, seg = (count + G - 1) / G;
      int rc;
      if (!staged) {
        rc = engine_score_device(e, tup + lo * W, m, dst + lo, s);
        if (rc) return cfail(c, rc, "%s", e->err);
        CHIP(c, hipEventRecord(c->ev_scored[b], s));
        CHIP(c, hipStreamWaitEvent(c->cs, c->ev_scored[b], 0));
        CNCCL(c, ncclAllReduce(dst + lo, dst + lo, m, ncclFloat, ncclSum, c->comm, c->cs));
      } else {
        if (c->slot_used[b]) CHIP(c, hipStreamWaitEvent(s, c->ev_free[b], 0));  // slot b still feeds chunk k-2's collective
        if (seg * G > count) CHIP(c, hipMemsetAsync(c->part[b] + count, 0, (seg * G - count) * sizeof(float), s));
        rc = K == 1 ? engine_score_device(e, tup + lo * W, m, c->part[b], s)
                    : engine_classify_device(e, tup + lo * W, m, c->part[b], nullptr, s);
        if (rc) return cfail(c, rc, "%s", e->err);
        CHIP(c, hipEventRecord(c->ev_scored[b], s));
        CHIP(c, hipStreamWaitEvent(c->cs, c->ev_scored[b], 0));
        const float* res;
        if (chain) {
          rc = chain_combine(c, b, count);
          if (rc) return rc;
          res = c->full[b];
        } else {
          CNCCL(c, ncclAllReduce(c->part[b], c->part[b], count, ncclFloat, ncclSum, c->comm, c->cs));
          res = c->part[b];
        }
        // [K][m] chunk block -> rows [lo, lo+m) of the [K][n_total] result
        CHIP(c, hipMemcpy2DAsync(dst + lo, n_total * sizeof(float), res, m * sizeof(float), m * sizeof(float), K, hipMemcpyDeviceToDevice, c->cs));
        CHIP(c, hipEventRecord(c->ev_free[b], c->cs));
        c->slot_used[b] = true;
      }
    }
    if (gather) {  // piece k of every row group changes hands (results are interleaved, not summed: ResultsCombiner.sv:371-391)
      GroupGuard grp;
      CNCCL(c, grp.start());
      const size_t mine = (size_t)c->row_group;
      for (size_t r = 0; r < Gr; ++r) {
        if (r == mine) continue;
        const int peer = (int)(r * G) + c->rank;  // the rank of row group r that holds my tree shard
        for (uint32_t kk = 0; kk < K; ++kk) {
          if (k < sched.size()) CNCCL(c, ncclSend(dst + (size_t)kk * n_total + lo, sched[k], ncclFloat, peer, c->world, c->cs));
          if (k < peer_sched[r].size())
            CNCCL(c, ncclRecv(dst + (size_t)kk * n_total + peer_lo[r] + peer_off[r], peer_sched[r][k], ncclFloat, peer, c->world, c->cs));
        }
      }
      CNCCL(c, grp.end());
      for (size_t r = 0; r < Gr; ++r)
        if (k < peer_sched[r].size()) peer_off[r] += peer_sched[r][k];
    }
    if (k < sched.size()) lo += sched[k];
  }
  CHIP(c, hipEventRecord(c->ev_done, c->cs));
  CHIP(c, hipStreamWaitEvent(s, c->ev_done, 0));  // results are ready in stream order on the caller's stream
  return DDT_OK;
}

static void count_sharded_job(ddt_comm* c, size_t scored, size_t returned) {
  c->e->st.score_calls++;
  c->e->st.tuples_in += scored;
  c->e->st.tuples_out += returned;
  c->e->st.tuple_lines_in += (uint64_t)scored * (tuple_words(c->e->p) / 4);
  c->e->st.result_lines_out += (returned + 3) / 4;
}

int ddt_score_sharded_device(ddt_comm* c, const void* d_tuples, size_t n, float* d_scores, int combine, void* stream) {
  int rc = check_call(c, d_tuples, d_scores, n);
  if (rc) return rc;
  if (c->e->num_classes != 1) return cfail(c, DDT_ESTATE, "multi-class model loaded: use ddt_classify_sharded_device");
  if (c->world) return cfail(c, DDT_ESTATE, "hybrid communicator: use ddt_score_hybrid_device");
  if (combine != DDT_COMBINE_ALLREDUCE && combine != DDT_COMBINE_CHAIN) return cfail(c, DDT_EINVAL, "combine %d", combine);
  if (n == 0) return DDT_OK;
  DeviceGuard dg(c->e->device);
  if (!dg.ok) return cfail(c, DDT_EHIP, "hipSetDevice(%d) failed", c->e->device);
  rc = sharded_pipeline(c, d_tuples, n, 0, n, d_scores, 1, combine, false, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  count_sharded_job(c, n, n);
  return DDT_OK;
}

// The hybrid job on device-resident buffers: this rank's row group scores rows [lo, hi) = ddt_hybrid_rows(n, row groups, row group)
// of the batch against the group's tree shards and combines them inside the group; gather != 0: every rank receives the other
// groups' rows too (all n scores everywhere, like the tree-sharded call), gather == 0: only [lo, hi) of d_scores is written (the
// reference returns a device's rows to the host from that device: ResultsCombiner.sv:371-391).  Only rows [lo, hi) of d_tuples are read.
int ddt_score_hybrid_device(ddt_comm* c, const void* d_tuples, size_t n, float* d_scores, int combine, int gather, void* stream) {
  int rc = check_call(c, d_tuples, d_scores, n);
  if (rc) return rc;
  if (c->e->num_classes != 1) return cfail(c, DDT_ESTATE, "multi-class model loaded: use ddt_classify_hybrid_device");
  if (!c->world) return cfail(c, DDT_ESTATE, "not a hybrid communicator (ddt_comm_create_hybrid)");
  if (combine != DDT_COMBINE_ALLREDUCE && combine != DDT_COMBINE_CHAIN) return cfail(c, DDT_EINVAL, "combine %d", combine);
  if (n == 0) return DDT_OK;
  DeviceGuard dg(c->e->device);
  if (!dg.ok) return cfail(c, DDT_EHIP, "hipSetDevice(%d) failed", c->e->device);
  size_t lo, hi;
  hybrid_window(n, (size_t)c->row_groups, (size_t)c->row_group, &lo, &hi);
  rc = sharded_pipeline(c, d_tuples, n, lo, hi, d_scores, 1, combine, gather != 0, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  count_sharded_job(c, hi - lo, gather ? n : hi - lo);
  return DDT_OK;
}

// m tuples of a host buffer onto THIS rank's device, asynchronous on stream s.  With peers the reference's scheme is kept: the
// tuples enter the machine once and travel on over the inter-device links (ring re-broadcast of InputDistributor.sv:199-204) --
// rank r copies rows [r * per, ...) over PCIe and every rank hands its rows to all peers with grouped ncclSend / ncclRecv
// straight into place (one message per xGMI link at once): 1/n of the PCIe and host-memory traffic per rank.  Collective.
static int tuples_to_device(ddt_comm* c, const uint32_t* src, size_t m, size_t W, uint32_t* d_tuples, hipStream_t s) {
  const size_t G = (size_t)c->n;
  const bool spread = (c->tuple_broadcast < 0 ? G > 1 : c->tuple_broadcast != 0) && G > 1;
  if (!spread) {
    CHIP(c, hipMemcpyAsync(d_tuples, src, m * W * 4, hipMemcpyHostToDevice, s));
    return DDT_OK;
  }
  const size_t per = (m + G - 1) / G, me = (size_t)c->rank;
  auto lo_of = [&](size_t r) { return std::min(r * per, m); };
  auto len_of = [&](size_t r) { return std::min(per, m - lo_of(r)); };
  if (len_of(me)) CHIP(c, hipMemcpyAsync(d_tuples + lo_of(me) * W, src + lo_of(me) * W, len_of(me) * W * 4, hipMemcpyHostToDevice, s));
  GroupGuard grp;
  CNCCL(c, grp.start());
  for (size_t r = 0; r < G; ++r) {
    if (r == me) continue;
    if (len_of(me)) CNCCL(c, ncclSend(d_tuples + lo_of(me) * W, len_of(me) * W, ncclFloat, (int)r, c->comm, s));  // 4-byte words
    if (len_of(r)) CNCCL(c, ncclRecv(d_tuples + lo_of(r) * W, len_of(r) * W, ncclFloat, (int)r, c->comm, s));
  }
  CNCCL(c, grp.end());
  return DDT_OK;
}

// Host-buffer form for one process per GPU (the per-rank counterpart of ddt_group_score): tuples from host memory to this
// rank's device in super-chunks of `host_rows`, the sharded job, the combined scores back to the host.  Collective: every
// rank calls it with the same tuples; every rank receives the scores.
int ddt_comm_score(ddt_comm* c, const void* tuple_lines, size_t n, float* scores_out, int combine) {
  if (!c) return DDT_EINVAL;
  if (c->dead) return cfail(c, DDT_ESTATE, "communicator was aborted");
  if (!c->e || !c->e->loaded) return cfail(c, DDT_ESTATE, "no model loaded on the engine of this communicator");
  if (c->e->num_classes != 1) return cfail(c, DDT_ESTATE, "multi-class model loaded");
  if (combine != DDT_COMBINE_ALLREDUCE && combine != DDT_COMBINE_CHAIN) return cfail(c, DDT_EINVAL, "combine %d", combine);
  if (n == 0) return DDT_OK;
  if (!tuple_lines || !scores_out) return cfail(c, DDT_EINVAL, "NULL host buffer");
  DeviceGuard dg(c->e->device);
  if (!dg.ok) return cfail(c, DDT_EHIP, "hipSetDevice(%d) failed", c->e->device);
  const size_t W = tuple_words(c->e->p), rows = std::min(std::min(c->host_rows, n), host_rows_cap(W));
  if (!c->hs) CHIP(c, hipStreamCreateWithFlags(&c->hs, hipStreamNonBlocking));
  if (rows > c->h_rows || W > c->h_words) {
    CHIP(c, hipStreamSynchronize(c->hs));
    if (c->h_tuples) (void)hipFree(c->h_tuples);
    if (c->h_scores) (void)hipFree(c->h_scores);
    c->h_tuples = nullptr;
    c->h_scores = nullptr;
    c->h_rows = c->h_words = 0;
    CHIP(c, hipMalloc(&c->h_tuples, rows * W * 4));
    CHIP(c, hipMalloc(reinterpret_cast<void**>(&c->h_scores), rows * sizeof(float)));
    c->h_rows = rows;
    c->h_words = W;
  }
  const uint32_t* src = reinterpret_cast<const uint32_t*>(tuple_lines);
  for (size_t off = 0; off < n; off += rows) {
    const size_t m = std::min(rows, n - off);
    int rc;
    if (c->world) {  // hybrid: only the row group's slice of the super-chunk comes to this device (1 / n of it over this rank's PCIe link)
      size_t lo, hi;
      hybrid_window(m, (size_t)c->row_groups, (size_t)c->row_group, &lo, &hi);
      rc = hi > lo ? tuples_to_device(c, src + (off + lo) * W, hi - lo, W, reinterpret_cast<uint32_t*>(c->h_tuples) + lo * W, c->hs) : DDT_OK;
      if (rc) return rc;
      rc = ddt_score_hybrid_device(c, c->h_tuples, m, c->h_scores, combine, 1, c->hs);
    } else {
      rc = tuples_to_device(c, src + off * W, m, W, reinterpret_cast<uint32_t*>(c->h_tuples), c->hs);
      if (rc) return rc;
      rc = ddt_score_sharded_device(c, c->h_tuples, m, c->h_scores, combine, c->hs);
    }
    if (rc) return rc;
    CHIP(c, hipMemcpyAsync(scores_out + off, c->h_scores, m * sizeof(float), hipMemcpyDeviceToHost, c->hs));
    CHIP(c, hipStreamSynchronize(c->hs));
  }
  return DDT_OK;
}

static int classify_job(ddt_comm* c, const void* d_tuples, size_t n, float* d_class_scores, int32_t* d_labels, int combine, bool hybrid, bool gather,
                        void* stream) {
  int rc = check_call(c, d_tuples, d_class_scores, n);
  if (rc) return rc;
  if (combine != DDT_COMBINE_ALLREDUCE && combine != DDT_COMBINE_CHAIN) return cfail(c, DDT_EINVAL, "combine %d", combine);
  if (n == 0) return DDT_OK;
  DeviceGuard dg(c->e->device);
  if (!dg.ok) return cfail(c, DDT_EHIP, "hipSetDevice(%d) failed", c->e->device);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const uint32_t K = c->e->num_classes;
  // K == 1 would take the in-place all-reduce path of the scalar scores: same result layout ([1][n])
  size_t lo = 0, hi = n;
  if (hybrid) hybrid_window(n, (size_t)c->row_groups, (size_t)c->row_group, &lo, &hi);
  rc = sharded_pipeline(c, d_tuples, n, lo, hi, d_class_scores, K, combine, hybrid && gather, s);
  if (rc) return rc;
  if (d_labels) {  // over the rows this rank holds combined sums of: all of them, or its row group's
    const bool all = !hybrid || gather || c->row_groups == 1;
    const size_t l0 = all ? 0 : lo, cnt = all ? n : hi - lo;
    hipError_t r = cnt ? launch_argmax_strided(d_class_scores + l0, K, n, cnt, d_labels + l0, s) : hipSuccess;
    if (r != hipSuccess) return cfail(c, DDT_EHIP, "argmax -> %s", hipGetErrorString(r));
  }
  c->e->st.score_calls++;
  c->e->st.tuples_in += hi - lo;
  c->e->st.tuples_out += (!hybrid || gather) ? n : hi - lo;
  return DDT_OK;
}

int ddt_classify_sharded_device(ddt_comm* c, const void* d_tuples, size_t n, float* d_class_scores, int32_t* d_labels, int combine,
                                void* stream) {
  if (c && c->world) return cfail(c, DDT_ESTATE, "hybrid communicator: use ddt_classify_hybrid_device");
  return classify_job(c, d_tuples, n, d_class_scores, d_labels, combine, false, false, stream);
}

// multi-class models through the hybrid job: per-class partial sums combined inside the row group, [K][n] layout as in
// ddt_classify_sharded_device; gather == 0: only the columns [lo, hi) of d_class_scores / d_labels are written
int ddt_classify_hybrid_device(ddt_comm* c, const void* d_tuples, size_t n, float* d_class_scores, int32_t* d_labels, int combine, int gather,
                               void* stream) {
  if (c && !c->world) return cfail(c, DDT_ESTATE, "not a hybrid communicator (ddt_comm_create_hybrid)");
  return classify_job(c, d_tuples, n, d_class_scores, d_labels, combine, true, gather != 0, stream);
}

int ddt_score_rowsharded_device(ddt_comm* c, const void* d_tuples, size_t n, float* d_scores, void* stream) {
  int rc = check_call(c, d_tuples, d_scores, n);
  if (rc) return rc;
  if (c->e->num_classes != 1) return cfail(c, DDT_ESTATE, "multi-class model loaded");
  if (c->world) return cfail(c, DDT_ESTATE, "hybrid communicator: use ddt_score_hybrid_device (tree_ranks 1 = replicas)");
  if (n == 0) return DDT_OK;
  DeviceGuard dg(c->e->device);
  if (!dg.ok) return cfail(c, DDT_EHIP, "hipSetDevice(%d) failed", c->e->device);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ddt_engine* e = c->e;
  const size_t W = tuple_words(e->p), G = (size_t)c->n, per = (n + G - 1) / G;
  // rank r owns rows [r * per, min((r + 1) * per, n)) and scores them IN PLACE in the caller's buffer, `step` rows at a time;
  // while step j + 1 is being scored the comm stream hands step j to every peer with grouped ncclSend / ncclRecv straight into
  // its place in their buffers (results are interleaved, not summed: ResultsCombiner.sv:371-391).  xGMI is a point-to-point
  // mesh: one message per link at once is how an all-gather uses it best, and exact per-peer counts need no padding or staging.
  auto len_of = [&](size_t r) { return std::min(per, n - std::min(r * per, n)); };
  size_t step = (std::min(c->chunk_rows, n) + G - 1) / G;
  if (step >= 1024) step = (step + 1023) / 1024 * 1024;  // whole tiles of the scoring kernels
  const size_t me = (size_t)c->rank, lo = std::min(me * per, n), mine = len_of(me);
  c->err[0] = 0;
  const uint32_t* tup = reinterpret_cast<const uint32_t*>(d_tuples);
  size_t k = 0;
  for (size_t off = 0; off < per; off += step, ++k) {
    const int b = (int)(k & 1);
    auto cnt = [&](size_t r) { return len_of(r) > off ? std::min(step, len_of(r) - off) : (size_t)0; };
    if (cnt(me)) {
      rc = engine_score_device(e, tup + (lo + off) * W, cnt(me), d_scores + lo + off, s);
      if (rc) return cfail(c, rc, "%s", e->err);
    }
    if (G == 1) continue;
    CHIP(c, hipEventRecord(c->ev_scored[b], s));
    CHIP(c, hipStreamWaitEvent(c->cs, c->ev_scored[b], 0));
    GroupGuard grp;
    CNCCL(c, grp.start());
    for (size_t r = 0; r < G; ++r) {
      if (r == me) continue;
      if (cnt(me)) CNCCL(c, ncclSend(d_scores + lo + off, cnt(me), ncclFloat, (int)r, c->comm, c->cs));
      if (cnt(r)) CNCCL(c, ncclRecv(d_scores + r * per + off, cnt(r), ncclFloat, (int)r, c->comm, c->cs));
    }
    CNCCL(c, grp.end());
  }
  if (G > 1) {
    CHIP(c, hipEventRecord(c->ev_done, c->cs));
    CHIP(c, hipStreamWaitEvent(s, c->ev_done, 0));  // every peer's rows have landed before the caller's stream moves on
  }
  e->st.score_calls++;
  e->st.tuples_in += mine;
  e->st.tuples_out += n;
  return DDT_OK;
}

}  // extern "C"

// =====================================================================================================
// single-process multi-GPU group
// =====================================================================================================
struct ddt_group {
  int n = 0;
  int tree_ranks = 0;  // 0 = plain (tree-sharded over all devices / replicas); Gt >= 1 = hybrid: row groups of Gt consecutive devices
  std::vector<int> devices;
  std::vector<ddt_engine*> eng;
  std::vector<ddt_comm*> comm;
  std::vector<hipStream_t> stream;
  std::vector<void*> d_tuples;
  std::vector<float*> d_scores;  // [cap_classes][cap_rows]: scalar scores, or the per-class sums of a multi-class model
  std::vector<int32_t*> d_labels;
  size_t cap_rows = 0, cap_words = 0, cap_classes = 0;
  size_t group_rows = 8u << 20;  // rows per super-chunk held on the devices at once
  char err[320] = {0};
};

namespace {

int gfail(ddt_group* g, int code, const char* fmt, ...) {
  if (g) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g->err, sizeof(g->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

// run fn(i) for every device on its own thread; first failing code wins
template <class F>
int for_each_device(ddt_group* g, F fn) {
  std::vector<int> rc((size_t)g->n, DDT_OK);
  std::vector<std::thread> th;
  for (int i = 1; i < g->n; ++i) th.emplace_back([&, i] { rc[(size_t)i] = fn(i); });
  rc[0] = fn(0);
  for (std::thread& t : th) t.join();
  for (int i = 0; i < g->n; ++i)
    if (rc[(size_t)i]) return rc[(size_t)i];
  return DDT_OK;
}

// tree shard of device i: i of n (plain), i % Gt of Gt (hybrid)
uint32_t shard_index(const ddt_group* g, int i) { return (uint32_t)(g->tree_ranks > 0 ? i % g->tree_ranks : i); }
uint32_t shard_count(const ddt_group* g) { return (uint32_t)(g->tree_ranks > 0 ? g->tree_ranks : g->n); }

void group_free_buffers(ddt_group* g) {
  for (int i = 0; i < g->n; ++i) {
    (void)hipSetDevice(g->devices[(size_t)i]);
    if (g->d_tuples[(size_t)i]) (void)hipFree(g->d_tuples[(size_t)i]);
    if (g->d_scores[(size_t)i]) (void)hipFree(g->d_scores[(size_t)i]);
    if (g->d_labels[(size_t)i]) (void)hipFree(g->d_labels[(size_t)i]);
    g->d_tuples[(size_t)i] = nullptr;
    g->d_scores[(size_t)i] = nullptr;
    g->d_labels[(size_t)i] = nullptr;
  }
  g->cap_rows = g->cap_words = g->cap_classes = 0;
}

}  // namespace

extern "C" {

static int group_create(ddt_group** out, int n_devices, const int* device_ids, int tree_ranks) {
  if (!out) return DDT_EINVAL;
  *out = nullptr;
  if (n_devices < 1 || n_devices > 64) return DDT_EINVAL;
  if (tree_ranks < 0 || (tree_ranks > 0 && n_devices % tree_ranks != 0)) return DDT_EINVAL;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return DDT_ENODEVICE;
  for (int i = 0; i < n_devices; ++i) {
    const int d = device_ids ? device_ids[i] : i;
    if (d < 0 || d >= count) return DDT_EINVAL;  // before anything is created
    for (int j = 0; j < i; ++j)
      if ((device_ids ? device_ids[j] : j) == d) return DDT_EINVAL;  // one communicator rank per device
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  std::unique_ptr<ddt_group> g(new (std::nothrow) ddt_group());
  if (!g) return DDT_ENOMEM;
  g->n = n_devices;
  g->tree_ranks = tree_ranks;
  for (int i = 0; i < n_devices; ++i) g->devices.push_back(device_ids ? device_ids[i] : i);
  g->eng.assign((size_t)n_devices, nullptr);
  g->comm.assign((size_t)n_devices, nullptr);
  g->stream.assign((size_t)n_devices, nullptr);
  g->d_tuples.assign((size_t)n_devices, nullptr);
  g->d_scores.assign((size_t)n_devices, nullptr);
  g->d_labels.assign((size_t)n_devices, nullptr);
  int rc = DDT_OK;
  for (int i = 0; i < n_devices && !rc; ++i) {
    rc = ddt_create(&g->eng[(size_t)i], g->devices[(size_t)i]);
    if (!rc && n_devices > 1) engine_enter_collective_job(g->eng[(size_t)i]);
  }
  std::vector<ncclComm_t> comms((size_t)n_devices, nullptr), subs((size_t)n_devices, nullptr);
  if (!rc && ncclCommInitAll(comms.data(), n_devices, g->devices.data()) != ncclSuccess) rc = DDT_EHIP;
  // hybrid: one more communicator per row group of tree_ranks consecutive devices (one process: ncclCommInitAll over the group's devices)
  for (int r0 = 0; tree_ranks > 0 && r0 < n_devices && !rc; r0 += tree_ranks)
    if (ncclCommInitAll(subs.data() + r0, tree_ranks, g->devices.data() + r0) != ncclSuccess) rc = DDT_EHIP;
  for (int i = 0; i < n_devices && !rc; ++i) {
    std::unique_ptr<ddt_comm> c(new (std::nothrow) ddt_comm());
    if (!c) {
      rc = DDT_ENOMEM;
      break;
    }
    c->e = g->eng[(size_t)i];
    c->device = g->devices[(size_t)i];
    c->rank = c->wrank = i;
    c->n = c->wn = n_devices;
    c->comm = comms[(size_t)i];
    comms[(size_t)i] = nullptr;
    if (tree_ranks > 0) {
      c->world = c->comm;
      c->comm = subs[(size_t)i];
      subs[(size_t)i] = nullptr;
      c->rank = i % tree_ranks;
      c->n = tree_ranks;
      c->row_groups = n_devices / tree_ranks;
      c->row_group = i / tree_ranks;
    }
    if (hipSetDevice(g->devices[(size_t)i]) != hipSuccess) rc = DDT_EHIP;
    if (!rc) rc = comm_init_common(c.get());
    if (!rc && hipStreamCreateWithFlags(&g->stream[(size_t)i], hipStreamNonBlocking) != hipSuccess) rc = DDT_EHIP;
    g->comm[(size_t)i] = c.release();
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  if (rc) {
    for (ncclComm_t c : comms)
      if (c) (void)ncclCommDestroy(c);
    for (ncclComm_t c : subs)
      if (c) (void)ncclCommDestroy(c);
    ddt_group_destroy(g.release());
    return rc;
  }
  *out = g.release();
  return DDT_OK;
}

int ddt_group_create(ddt_group** out, int n_devices, const int* device_ids) { return group_create(out, n_devices, device_ids, 0); }

// the hybrid layout in one process: row groups of `tree_ranks` consecutive devices; ddt_group_load_model* then gives device i tree shard
// i % tree_ranks of tree_ranks, ddt_group_score / ddt_group_classify run the hybrid job and return all rows from device 0
int ddt_group_create_hybrid(ddt_group** out, int n_devices, const int* device_ids, int tree_ranks) {
  if (tree_ranks < 1) return DDT_EINVAL;
  return group_create(out, n_devices, device_ids, tree_ranks);
}

void ddt_group_destroy(ddt_group* g) {
  if (!g) return;
  int prev = -1;
  (void)hipGetDevice(&prev);
  group_free_buffers(g);
  for (int i = 0; i < g->n; ++i) {
    (void)hipSetDevice(g->devices[(size_t)i]);
    if (g->stream[(size_t)i]) {
      (void)hipStreamSynchronize(g->stream[(size_t)i]);
      (void)hipStreamDestroy(g->stream[(size_t)i]);
    }
    if (g->comm[(size_t)i]) ddt_comm_destroy(g->comm[(size_t)i]);
    if (g->eng[(size_t)i]) ddt_destroy(g->eng[(size_t)i]);
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  delete g;
}

const char* ddt_group_last_error(const ddt_group* g) { return g ? g->err : ""; }

ddt_engine* ddt_group_engine(ddt_group* g, int index) { return (g && index >= 0 && index < g->n) ? g->eng[(size_t)index] : nullptr; }

int ddt_group_load_model(ddt_group* g, const ddt_params* p, const void* wl, size_t n_wlines, const void* fl, size_t n_flines) {
  if (!g) return DDT_EINVAL;
  int rc = for_each_device(g, [&](int i) { return ddt_load_model_shard(g->eng[(size_t)i], p, wl, n_wlines, fl, n_flines, shard_index(g, i), shard_count(g)); });
  if (rc)
    for (int i = 0; i < g->n; ++i)
      if (g->eng[(size_t)i]->err[0]) return gfail(g, rc, "device %d: %s", g->devices[(size_t)i], g->eng[(size_t)i]->err);
  return rc;
}

int ddt_group_load_model_sparse(ddt_group* g, const ddt_params* p, const void* node_lines, size_t n_lines, const uint64_t* first) {
  if (!g) return DDT_EINVAL;
  int rc = for_each_device(g, [&](int i) { return ddt_load_model_sparse(g->eng[(size_t)i], p, node_lines, n_lines, first, shard_index(g, i), shard_count(g)); });
  if (rc)
    for (int i = 0; i < g->n; ++i)
      if (g->eng[(size_t)i]->err[0]) return gfail(g, rc, "device %d: %s", g->devices[(size_t)i], g->eng[(size_t)i]->err);
  return rc;
}

int ddt_group_load_model_multiclass(ddt_group* g, const ddt_params* p, const void* wl, size_t n_wlines, const void* fl, size_t n_flines,
                                    uint32_t num_classes, int interleaved) {
  if (!g) return DDT_EINVAL;
  int rc = for_each_device(g, [&](int i) {
    return ddt_load_model_multiclass(g->eng[(size_t)i], p, wl, n_wlines, fl, n_flines, num_classes, interleaved, shard_index(g, i), shard_count(g));
  });
  if (rc)
    for (int i = 0; i < g->n; ++i)
      if (g->eng[(size_t)i]->err[0]) return gfail(g, rc, "device %d: %s", g->devices[(size_t)i], g->eng[(size_t)i]->err);
  return rc;
}

}  // extern "C"

namespace {

// host tuples -> every device -> the sharded job -> device 0's results back to the host, in super-chunks of group_rows.
// classes == 1: scores_out [n].  classes > 1: labels_out [n] and (optional) class_scores_out [classes][n].
int group_run(ddt_group* g, const void* tuple_lines, size_t n, float* scores_out, int32_t* labels_out, float* class_scores_out, int combine) {
  if (!g->eng[0]->loaded) return gfail(g, DDT_ESTATE, "no model loaded");
  const size_t K = g->eng[0]->num_classes;
  const bool classify = labels_out != nullptr || class_scores_out != nullptr;
  if (classify != (K > 1)) return gfail(g, DDT_ESTATE, K > 1 ? "multi-class model loaded: use ddt_group_classify" : "scalar model loaded: use ddt_group_score");
  const size_t W = tuple_words(g->eng[0]->p);
  const size_t rows = std::min(std::min(g->group_rows, n), host_rows_cap(W));
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (rows > g->cap_rows || W > g->cap_words || K > g->cap_classes) {
    group_free_buffers(g);
    for (int i = 0; i < g->n; ++i) {
      if (hipSetDevice(g->devices[(size_t)i]) != hipSuccess || hipMalloc(&g->d_tuples[(size_t)i], rows * W * 4) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&g->d_scores[(size_t)i]), K * rows * sizeof(float)) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&g->d_labels[(size_t)i]), rows * sizeof(int32_t)) != hipSuccess) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return gfail(g, DDT_ENOMEM, "device %d: tuple / score buffers for %zu rows", g->devices[(size_t)i], rows);
      }
    }
    g->cap_rows = rows;
    g->cap_words = W;
    g->cap_classes = K;
  }
  const uint32_t* src = reinterpret_cast<const uint32_t*>(tuple_lines);
  int rc = DDT_OK;
  for (size_t off = 0; off < n && !rc; off += rows) {
    const size_t m = std::min(rows, n - off);
    // every device: its own copy of the tuples (the reference broadcasts them along the ring), the sharded job, and
    // -- device 0 only -- the combined results back to the host
    rc = for_each_device(g, [&](int i) -> int {
      const size_t k = (size_t)i;
      hipStream_t s = g->stream[k];
      if (hipSetDevice(g->devices[k]) != hipSuccess) return DDT_EHIP;
      int r;
      if (g->tree_ranks > 0) {  // hybrid: the row group's slice of the super-chunk only, then all rows gathered (device 0 returns them)
        size_t lo, hi;
        hybrid_window(m, (size_t)g->comm[k]->row_groups, (size_t)g->comm[k]->row_group, &lo, &hi);
        if (hi > lo && tuples_to_device(g->comm[k], src + (off + lo) * W, hi - lo, W, reinterpret_cast<uint32_t*>(g->d_tuples[k]) + lo * W, s)) return DDT_EHIP;
        r = classify ? ddt_classify_hybrid_device(g->comm[k], g->d_tuples[k], m, g->d_scores[k], g->d_labels[k], combine, 1, s)
                     : ddt_score_hybrid_device(g->comm[k], g->d_tuples[k], m, g->d_scores[k], combine, 1, s);
      } else {
        if (tuples_to_device(g->comm[k], src + off * W, m, W, reinterpret_cast<uint32_t*>(g->d_tuples[k]), s)) return DDT_EHIP;
        r = classify ? ddt_classify_sharded_device(g->comm[k], g->d_tuples[k], m, g->d_scores[k], g->d_labels[k], combine, s)
                     : ddt_score_sharded_device(g->comm[k], g->d_tuples[k], m, g->d_scores[k], combine, s);
      }
      if (r) return r;
      if (i == 0) {
        bool ok = true;
        if (!classify) ok = hipMemcpyAsync(scores_out + off, g->d_scores[k], m * sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess;
        if (classify && labels_out) ok = hipMemcpyAsync(labels_out + off, g->d_labels[k], m * sizeof(int32_t), hipMemcpyDeviceToHost, s) == hipSuccess;
        for (size_t c = 0; classify && class_scores_out && ok && c < K; ++c)  // device layout [K][m] of this super-chunk -> host [K][n]
          ok = hipMemcpyAsync(class_scores_out + c * n + off, g->d_scores[k] + c * m, m * sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess;
        if (!ok) return DDT_EHIP;
      }
      return hipStreamSynchronize(s) == hipSuccess ? DDT_OK : DDT_EHIP;
    });
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  if (rc)
    for (int i = 0; i < g->n; ++i) {
      const char* msg = ddt_comm_last_error(g->comm[(size_t)i]);
      if (msg && msg[0]) return gfail(g, rc, "device %d: %s", g->devices[(size_t)i], msg);
    }
  return rc;
}

}  // namespace

extern "C" {

int ddt_group_score(ddt_group* g, const void* tuple_lines, size_t n, float* scores_out, int combine) {
  if (!g) return DDT_EINVAL;
  if (n == 0) return DDT_OK;
  if (!tuple_lines || !scores_out) return gfail(g, DDT_EINVAL, "NULL host buffer");
  return group_run(g, tuple_lines, n, scores_out, nullptr, nullptr, combine);
}

// The reference's other mode for a single process: every device holds the WHOLE ensemble (DTInference.sv:33-36 "trees
// broadcast") and scores its share of the rows through its own engine's feeder -- no collective, G PCIe links in parallel.
int ddt_group_load_model_replicated(ddt_group* g, const ddt_params* p, const void* wl, size_t n_wlines, const void* fl, size_t n_flines) {
  if (!g) return DDT_EINVAL;
  int rc = for_each_device(g, [&](int i) { return ddt_load_model(g->eng[(size_t)i], p, wl, n_wlines, fl, n_flines); });
  if (rc)
    for (int i = 0; i < g->n; ++i)
      if (g->eng[(size_t)i]->err[0]) return gfail(g, rc, "device %d: %s", g->devices[(size_t)i], g->eng[(size_t)i]->err);
  return rc;
}

int ddt_group_score_rows(ddt_group* g, const void* tuple_lines, size_t n, float* scores_out) {
  if (!g) return DDT_EINVAL;
  if (n == 0) return DDT_OK;
  if (!tuple_lines || !scores_out) return gfail(g, DDT_EINVAL, "NULL host buffer");
  for (int i = 0; i < g->n; ++i) {
    const ddt_engine* e = g->eng[(size_t)i];
    if (!e->loaded || e->num_classes != 1) return gfail(g, DDT_ESTATE, "device %d: no scalar model loaded", g->devices[(size_t)i]);
    ddt_info info;
    if (ddt_get_info(e, &info) || info.tree_begin != 0 || info.tree_end != e->p.num_trees)
      return gfail(g, DDT_ESTATE, "device %d holds a tree shard: row partitioning needs ddt_group_load_model_replicated", g->devices[(size_t)i]);
  }
  const size_t W = tuple_words(g->eng[0]->p), G = (size_t)g->n, per = ((n + G - 1) / G + 3) / 4 * 4;  // whole result lines per device
  const uint32_t* src = reinterpret_cast<const uint32_t*>(tuple_lines);
  int rc = for_each_device(g, [&](int i) -> int {
    const size_t lo = std::min((size_t)i * per, n), hi = std::min(lo + per, n);
    return hi > lo ? ddt_score(g->eng[(size_t)i], src + lo * W, hi - lo, scores_out + lo) : DDT_OK;
  });
  if (rc)
    for (int i = 0; i < g->n; ++i)
      if (g->eng[(size_t)i]->err[0]) return gfail(g, rc, "device %d: %s", g->devices[(size_t)i], g->eng[(size_t)i]->err);
  return rc;
}

int ddt_group_classify(ddt_group* g, const void* tuple_lines, size_t n, int32_t* labels_out, float* class_scores_out, int combine) {
  if (!g) return DDT_EINVAL;
  if (n == 0) return DDT_OK;
  if (!tuple_lines || !labels_out) return gfail(g, DDT_EINVAL, "NULL host buffer");
  return group_run(g, tuple_lines, n, nullptr, labels_out, class_scores_out, combine);
}

}  // extern "C"
