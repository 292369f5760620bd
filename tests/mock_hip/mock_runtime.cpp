// mock_runtime.cpp -- TEST INFRASTRUCTURE: a deferred-execution model of the HIP streams / events and RCCL collectives that
// csrc/ddt_comm.cpp uses, plus a stand-in engine, so that the REAL multi-GPU pipeline code (chunk schedule, workspace slots,
// event protocol, all-to-all segment arithmetic, [K][n] layouts, row partitions, the single-process group) runs with
// G = 2 .. 8 ranks on a machine without GPUs (tests/test_comm_mock.py builds this file together with csrc/ddt_comm.cpp).
//
// Model: every stream is a FIFO of operations that are NOT executed when enqueued.  A scheduler executes one runnable
// operation at a time -- runnable = at the head of its stream and every event it waits for has been recorded -- choosing
// among the candidates by a policy: lowest stream id first, highest first, or seeded random.  Only what HIP guarantees is
// honoured (stream order, hipStreamWaitEvent on the record that was enqueued last, blocking synchronisation calls), so a
// missing dependency in the pipeline gives a wrong result under some schedule.  A collective executes when all ranks'
// copies are runnable (that is RCCL's rendezvous).  "Device memory" is host memory; kernels are host functions.
//
// This file is #included by the two translation units that complete a build: mock_engine.cpp (a stand-in engine with
// integer-valued partial scores: the communicator tests) or mock_kernels.cpp (CPU stand-ins for the kernels that read the REAL
// packed images, under the real csrc/ddt_engine.cpp: the engine tests).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <random>
#include <set>
#include <vector>

#include "ddt_engine_priv.h"

namespace {

std::mutex M;
std::condition_variable CV;

struct Coll;
struct Op {
  std::function<void()> run;
  std::vector<uint64_t> deps;
  uint64_t record = 0;
  Coll* coll = nullptr;
  double cost = 0.0;  // model time units (see "timeline" below)
};
}  // namespace

struct MockStream {
  int dev = 0, id = 0;
  std::deque<Op*> q;
  uint64_t enq = 0, done = 0;
  double t = 0.0;  // timeline: when the stream's last executed operation ended
};
struct MockEvent {
  uint64_t serial = 0;  // last record enqueued on this event (0 = never recorded: a wait is a no-op, as in HIP)
};

namespace {
struct Group;
}
struct MockComm {
  Group* g = nullptr;
  int rank = 0;
  uint64_t next_seq = 0, next_split = 0;
};

namespace {

struct P2P {
  bool send;
  int peer;
  const void* sbuf;
  void* rbuf;
  size_t count;
};
struct CollArg {
  int rank = -1;  // of the communicator
  int type = -1;  // 0 all-reduce, 1 all-gather, 2 grouped send / recv
  const void* send = nullptr;
  void* recv = nullptr;
  size_t count = 0;
  std::vector<P2P> p2p;
  Op* op = nullptr;
  MockStream* st = nullptr;
};
struct Coll {
  Group* g;
  uint64_t seq;
  int arrived = 0;
  std::vector<CollArg> arg;
};
struct Split {  // one ncclCommSplit call of a communicator: every rank brings (colour, key), the last one builds the new groups
  int arrived = 0;
  std::vector<int> color, key;
  std::vector<MockComm*> out;
  bool done = false;
};
struct Group {
  int n = 0, joined = 0;
  std::vector<MockComm*> comm;
  std::map<uint64_t, Coll*> pending;
  std::map<uint64_t, Split> splits;
  std::vector<char> aborted;  // per rank: ncclCommAbort was called -- that rank's queued collectives end without running
};

std::set<uint64_t> g_done;
std::map<uint64_t, double> g_done_at;  // timeline: when an event record completed
// Timeline: besides executing the operations the model keeps an as-soon-as-possible clock -- an operation starts when its
// stream is free and its event dependencies (for a collective: those of every rank) have completed, and lasts `cost` units:
// scoring kernels g_cost_row per row, an all-reduce / all-gather g_cost_float per float of its result, grouped point-to-point
// messages g_cost_float per float of the LONGEST message (one message per link at once), copies g_cost_copy per float.
// mock_makespan() = the latest end over all streams: what the pipeline's overlap structure makes of those costs.
double g_cost_row = 0.0, g_cost_float = 0.0, g_cost_copy = 0.0;
uint64_t g_serial = 1, g_executed = 0, g_h2d_bytes = 0;  // g_h2d_bytes: what crossed "PCIe" towards the devices
std::vector<MockStream*> g_streams;
std::map<int, MockStream*> g_null;
std::map<std::string, Group*> g_groups;  // by unique id
int g_policy = 0, g_devices = 8, g_errors = 0, g_next_stream = 1;
uint64_t g_next_id = 1;
std::mt19937_64 g_rng(1);
thread_local int t_dev = 0;
thread_local bool t_in_group = false;
thread_local std::vector<P2P> t_p2p;
thread_local MockComm* t_group_comm = nullptr;
thread_local MockStream* t_group_stream = nullptr;

MockStream* resolve(hipStream_t s) {  // M held
  if (s) return s;
  MockStream*& d = g_null[t_dev];
  if (!d) {
    d = new MockStream();
    d->dev = t_dev;
    d->id = 0;
    g_streams.push_back(d);
  }
  return d;
}

bool deps_done(const Op* op) {
  for (uint64_t d : op->deps)
    if (!g_done.count(d)) return false;
  return true;
}

bool rank_aborted(const Group* g, int rank) { return rank >= 0 && (size_t)rank < g->aborted.size() && g->aborted[(size_t)rank]; }

// the rank of the communicator whose copy of collective c is operation op
int coll_rank_of(const Coll* c, const Op* op) {
  for (const CollArg& a : c->arg)
    if (a.op == op) return a.rank;
  return -1;
}

bool coll_ready(const Coll* c) {
  if (c->arrived != c->g->n) return false;
  for (const CollArg& a : c->arg)
    if (!a.st || !a.op || a.st->q.empty() || a.st->q.front() != a.op || !deps_done(a.op)) return false;  // (a rank that aborted: never)
  return true;
}

void run_coll(Coll* c) {
  const int n = c->g->n;
  const CollArg& a0 = c->arg[0];
  for (const CollArg& a : c->arg)
    if (a.type != a0.type || (a.type != 2 && a.count != a0.count)) {
      fprintf(stderr, "[mock] collective %llu: ranks disagree (type %d/%d, count %zu/%zu)\n", (unsigned long long)c->seq, a.type, a0.type, a.count,
              a0.count);
      ++g_errors;
      return;
    }
  if (a0.type == 0) {  // sum in rank order (the partials are integer-valued: any order gives the same bits)
    std::vector<float> acc(a0.count, 0.0f);
    for (int r = 0; r < n; ++r) {
      const float* s = reinterpret_cast<const float*>(c->arg[(size_t)r].send);
      for (size_t i = 0; i < a0.count; ++i) {
        volatile float v = acc[i] + s[i];
        acc[i] = v;
      }
    }
    for (int r = 0; r < n; ++r) memcpy(c->arg[(size_t)r].recv, acc.data(), a0.count * sizeof(float));
  } else if (a0.type == 1) {
    std::vector<float> all((size_t)n * a0.count);
    for (int r = 0; r < n; ++r) memcpy(all.data() + (size_t)r * a0.count, c->arg[(size_t)r].send, a0.count * sizeof(float));
    for (int r = 0; r < n; ++r) memcpy(c->arg[(size_t)r].recv, all.data(), all.size() * sizeof(float));
  } else {  // grouped send / recv: the k-th send of rank r to peer d pairs with the k-th recv of rank d from peer r
    std::vector<std::vector<float>> staged;  // read everything first: buffers may alias
    std::vector<std::pair<void*, size_t>> dst;
    for (int r = 0; r < n; ++r) {
      std::map<int, int> nth;
      for (const P2P& p : c->arg[(size_t)r].p2p) {
        if (!p.send) continue;
        const int k = nth[p.peer]++;
        int seen = 0;
        const P2P* match = nullptr;
        for (const P2P& q : c->arg[(size_t)p.peer].p2p)
          if (!q.send && q.peer == r && seen++ == k) {
            match = &q;
            break;
          }
        if (!match || match->count != p.count) {
          fprintf(stderr, "[mock] send %d -> %d (%zu floats) has no matching recv\n", r, p.peer, p.count);
          ++g_errors;
          continue;
        }
        staged.emplace_back(reinterpret_cast<const float*>(p.sbuf), reinterpret_cast<const float*>(p.sbuf) + p.count);
        dst.emplace_back(match->rbuf, p.count);
      }
    }
    size_t recvs = 0;
    for (int r = 0; r < n; ++r)
      for (const P2P& p : c->arg[(size_t)r].p2p) recvs += p.send ? 0 : 1;
    if (recvs != dst.size()) {
      fprintf(stderr, "[mock] %zu recvs for %zu sends\n", recvs, dst.size());
      ++g_errors;
    }
    for (size_t i = 0; i < dst.size(); ++i) memcpy(dst[i].first, staged[i].data(), dst[i].second * sizeof(float));
  }
}

double deps_time(const Op* op) {
  double t = 0.0;
  for (uint64_t d : op->deps) t = std::max(t, g_done_at[d]);
  return t;
}

void finish(MockStream* st, Op* op) {
  st->q.pop_front();
  ++st->done;
  if (!op->coll) st->t = std::max(st->t, deps_time(op)) + op->cost;  // collectives: set for all ranks in step()
  if (op->record) {
    g_done.insert(op->record);
    g_done_at[op->record] = st->t;
  }
  delete op;
  ++g_executed;
}

// one scheduling step; returns false when nothing is runnable.  M held.
bool step() {
  std::vector<MockStream*> cand;
  for (MockStream* st : g_streams) {
    if (st->q.empty()) continue;
    Op* op = st->q.front();
    if (op->coll && rank_aborted(op->coll->g, coll_rank_of(op->coll, op)) && deps_done(op)) cand.push_back(st);  // ends alone, without running
    else if (op->coll ? coll_ready(op->coll) : deps_done(op)) cand.push_back(st);
  }
  if (cand.empty()) return false;
  MockStream* st = cand[0];
  if (g_policy == 1) {
    for (MockStream* c : cand)
      if (c->id > st->id) st = c;
  } else if (g_policy == 2) {
    st = cand[g_rng() % cand.size()];
  } else {
    for (MockStream* c : cand)
      if (c->id < st->id) st = c;
  }
  Op* op = st->q.front();
  if (op->coll && rank_aborted(op->coll->g, coll_rank_of(op->coll, op))) {
    for (CollArg& a : op->coll->arg)
      if (a.op == op) a.op = nullptr, a.st = nullptr;  // (the Coll object is left to the process: peers may still point at it)
    op->coll = nullptr;
    finish(st, op);
  } else if (op->coll) {
    Coll* c = op->coll;
    run_coll(c);
    double start = 0.0, cost = 0.0;
    for (CollArg& a : c->arg) {
      start = std::max(start, std::max(a.st->t, deps_time(a.op)));
      size_t floats = a.count * (a.type == 1 ? (size_t)c->g->n : 1u);
      const int self = (int)(&a - c->arg.data());
      for (const P2P& p : a.p2p)  // point-to-point messages to different peers travel on different links: the longest one counts
        if (p.send && p.peer != self && p.count > floats) floats = p.count;
      cost = std::max(cost, (double)floats * g_cost_float);
    }
    for (CollArg& a : c->arg) a.st->t = start + cost;
    c->g->pending.erase(c->seq);
    for (CollArg& a : c->arg) finish(a.st, a.op);
    delete c;
  } else {
    if (op->run) op->run();
    finish(st, op);
  }
  return true;
}

template <class Pred>
hipError_t drive(std::unique_lock<std::mutex>& lk, Pred pred) {
  while (!pred()) {
    if (step()) {
      CV.notify_all();
      continue;
    }
    if (CV.wait_for(lk, std::chrono::seconds(20)) == std::cv_status::timeout && !pred()) {
      fprintf(stderr, "[mock] no progress for 20 s: a rank is waiting for work that nobody will enqueue\n");
      ++g_errors;
      return hipErrorInvalidValue;
    }
  }
  return hipSuccess;
}

void enqueue(hipStream_t s, Op* op) {
  std::lock_guard<std::mutex> lk(M);
  MockStream* st = resolve(s);
  st->q.push_back(op);
  ++st->enq;
  CV.notify_all();
}

ncclResult_t enqueue_coll(MockComm* c, hipStream_t s, CollArg a) {
  std::lock_guard<std::mutex> lk(M);
  MockStream* st = resolve(s);
  Group* g = c->g;
  const uint64_t seq = c->next_seq++;
  Coll*& k = g->pending[seq];
  if (!k) {
    k = new Coll{g, seq, 0, std::vector<CollArg>((size_t)g->n)};
  }
  Op* op = new Op();
  op->coll = k;
  a.op = op;
  a.st = st;
  a.rank = c->rank;
  k->arg[(size_t)c->rank] = a;
  ++k->arrived;
  st->q.push_back(op);
  ++st->enq;
  CV.notify_all();
  return ncclSuccess;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------ HIP
extern "C" {

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "mock hip error"; }
hipError_t hipGetDeviceCount(int* n) {
  *n = g_devices;
  return hipSuccess;
}
hipError_t hipGetDevice(int* d) {
  *d = t_dev;
  return hipSuccess;
}
hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= g_devices) return hipErrorInvalidDevice;
  t_dev = d;
  return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t bytes) {
  *p = malloc(bytes ? bytes : 1);
  if (*p) memset(*p, 0xA5, bytes);  // never zero: stale reads show
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
  (void)hipDeviceSynchronize();  // hipFree waits for the device
  free(p);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  std::lock_guard<std::mutex> lk(M);
  MockStream* st = new MockStream();
  st->dev = t_dev;
  st->id = g_next_stream++;
  g_streams.push_back(st);
  *s = st;
  return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) {
  *least = 0;
  *greatest = -1;
  return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }  // order only
hipError_t hipStreamSynchronize(hipStream_t s) {
  std::unique_lock<std::mutex> lk(M);
  MockStream* st = resolve(s);
  const uint64_t target = st->enq;
  return drive(lk, [&] { return st->done >= target; });
}
hipError_t hipStreamDestroy(hipStream_t s) {
  if (!s) return hipErrorInvalidValue;
  (void)hipStreamSynchronize(s);
  std::lock_guard<std::mutex> lk(M);
  for (size_t i = 0; i < g_streams.size(); ++i)
    if (g_streams[i] == s) g_streams.erase(g_streams.begin() + (long)i);
  delete s;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) {
  std::unique_lock<std::mutex> lk(M);
  std::vector<std::pair<MockStream*, uint64_t>> targets;
  for (MockStream* st : g_streams)
    if (st->dev == t_dev) targets.emplace_back(st, st->enq);
  return drive(lk, [&] {
    for (auto& t : targets)
      if (t.first->done < t.second) return false;
    return true;
  });
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
  *e = new MockEvent();
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  Op* op = new Op();
  {
    std::lock_guard<std::mutex> lk(M);
    op->record = g_serial++;
    e->serial = op->record;
  }
  enqueue(s, op);
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  Op* op = new Op();
  {
    std::lock_guard<std::mutex> lk(M);
    if (e->serial) op->deps.push_back(e->serial);  // the record enqueued last, as HIP defines it
  }
  enqueue(s, op);
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t attr, int device) {
  if (device < 0 || device >= g_devices) return hipErrorInvalidDevice;
  if (attr != hipDeviceAttributeWallClockRate) return hipErrorInvalidValue;
  *value = 100000;  // kHz: the constant 100 MHz clock s_memrealtime reads
  return hipSuccess;
}

hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int device) {
  if (device < 0 || device >= g_devices) return hipErrorInvalidDevice;
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "mock device %d", device);
  snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:mock");
  p->multiProcessorCount = 256;
  p->clockRate = 2400000;
  p->sharedMemPerBlock = p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
  *p = malloc(bytes ? bytes : 1);
  if (*p) memset(*p, 0x5A, bytes);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) {
  free(p);
  return hipSuccess;
}
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventSynchronize(hipEvent_t e) {
  std::unique_lock<std::mutex> lk(M);
  const uint64_t target = e->serial;
  if (!target) return hipSuccess;
  return drive(lk, [&] { return g_done.count(target) != 0; });
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  std::lock_guard<std::mutex> lk(M);
  *ms = (float)(g_done_at[b->serial] - g_done_at[a->serial]);  // model time units
  return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
  (void)hipDeviceSynchronize();  // the blocking copy is ordered behind everything the device has been given
  memmove(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t s) {
  Op* op = new Op();
  op->run = [=] { memset(p, value, bytes); };
  enqueue(s, op);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
  if (kind == hipMemcpyHostToDevice) {
    std::lock_guard<std::mutex> lk(M);
    g_h2d_bytes += bytes;
  }
  Op* op = new Op();
  op->run = [=] { memmove(dst, src, bytes); };
  enqueue(s, op);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t s) {
  Op* op = new Op();
  op->cost = (double)(width / 4 * height) * g_cost_copy;
  op->run = [=] {
    for (size_t r = 0; r < height; ++r) memmove(reinterpret_cast<char*>(dst) + r * dpitch, reinterpret_cast<const char*>(src) + r * spitch, width);
  };
  enqueue(s, op);
  return hipSuccess;
}

// ----------------------------------------------------------------------------------------------------------------- RCCL
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "mock nccl error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::lock_guard<std::mutex> lk(M);
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "mock-id-%llu", (unsigned long long)g_next_id++);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int n, ncclUniqueId id, int rank) {
  std::unique_lock<std::mutex> lk(M);
  Group*& g = g_groups[std::string(id.internal)];
  if (!g) {
    g = new Group();
    g->n = n;
    g->comm.assign((size_t)n, nullptr);
    g->aborted.assign((size_t)n, 0);
  }
  if (g->n != n || rank < 0 || rank >= n || g->comm[(size_t)rank]) return ncclInvalidArgument;
  MockComm* c = new MockComm();
  c->g = g;
  c->rank = rank;
  g->comm[(size_t)rank] = c;
  ++g->joined;
  CV.notify_all();
  Group* gg = g;
  while (gg->joined < gg->n)  // ncclCommInitRank returns when every rank has joined
    if (CV.wait_for(lk, std::chrono::seconds(20)) == std::cv_status::timeout) return ncclInternalError;
  *comm = c;
  return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) {
  std::lock_guard<std::mutex> lk(M);
  Group* g = new Group();
  g->n = g->joined = n;
  g->aborted.assign((size_t)n, 0);
  for (int r = 0; r < n; ++r) {
    MockComm* c = new MockComm();
    c->g = g;
    c->rank = r;
    g->comm.push_back(c);
    comms[r] = c;
  }
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete comm;  // the group object is left to the process (tests are short lived)
  return ncclSuccess;
}
// every rank of `comm` calls it (a rendezvous, like ncclCommInitRank); ranks of one colour form a new communicator, ordered by key
ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, ncclConfig_t*) {
  std::unique_lock<std::mutex> lk(M);
  Group* g = comm->g;
  const uint64_t idx = comm->next_split++;
  Split& sp = g->splits[idx];
  if (sp.color.empty()) {
    sp.color.assign((size_t)g->n, -2);
    sp.key.assign((size_t)g->n, 0);
    sp.out.assign((size_t)g->n, nullptr);
  }
  sp.color[(size_t)comm->rank] = color;
  sp.key[(size_t)comm->rank] = key;
  if (++sp.arrived == g->n) {
    std::map<int, std::vector<std::pair<int, int>>> by_color;  // colour -> (key, parent rank)
    for (int r = 0; r < g->n; ++r)
      if (sp.color[(size_t)r] >= 0) by_color[sp.color[(size_t)r]].emplace_back(sp.key[(size_t)r], r);
    for (auto& kv : by_color) {
      std::sort(kv.second.begin(), kv.second.end());
      Group* ng = new Group();
      ng->n = ng->joined = (int)kv.second.size();
      ng->aborted.assign(kv.second.size(), 0);
      for (size_t i = 0; i < kv.second.size(); ++i) {
        MockComm* c = new MockComm();
        c->g = ng;
        c->rank = (int)i;
        ng->comm.push_back(c);
        sp.out[(size_t)kv.second[i].second] = c;
      }
    }
    sp.done = true;
    CV.notify_all();
  }
  while (!sp.done)
    if (CV.wait_for(lk, std::chrono::seconds(20)) == std::cv_status::timeout) return ncclInternalError;
  *newcomm = sp.out[(size_t)comm->rank];
  return ncclSuccess;
}
// local, like RCCL's: THIS rank's queued collectives of the communicator end without running (its streams drain); the peers' copies
// keep waiting for a rendezvous that will not come until they abort too
ncclResult_t ncclCommAbort(ncclComm_t comm) {
  std::lock_guard<std::mutex> lk(M);
  comm->g->aborted[(size_t)comm->rank] = 1;
  CV.notify_all();
  return ncclSuccess;  // (the MockComm is left to the process: queued operations still point at its group)
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t, ncclRedOp_t, ncclComm_t comm, hipStream_t s) {
  CollArg a;
  a.type = 0, a.send = send, a.recv = recv, a.count = count;
  return enqueue_coll(comm, s, a);
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t, ncclComm_t comm, hipStream_t s) {
  CollArg a;
  a.type = 1, a.send = send, a.recv = recv, a.count = sendcount;
  return enqueue_coll(comm, s, a);
}
ncclResult_t ncclGroupStart(void) {
  t_in_group = true;
  t_p2p.clear();
  t_group_comm = nullptr;
  t_group_stream = nullptr;
  return ncclSuccess;
}
static ncclResult_t p2p(bool send, const void* sbuf, void* rbuf, size_t count, int peer, ncclComm_t comm, hipStream_t s) {
  if (!t_in_group) return ncclInvalidArgument;  // the pipeline only issues grouped point-to-point calls
  if ((t_group_comm && t_group_comm != comm) || (t_group_comm && t_group_stream != s)) return ncclInvalidArgument;
  t_group_comm = comm;
  t_group_stream = s;
  t_p2p.push_back(P2P{send, peer, sbuf, rbuf, count});
  return ncclSuccess;
}
ncclResult_t ncclSend(const void* send, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t s) {
  return p2p(true, send, nullptr, count, peer, comm, s);
}
ncclResult_t ncclRecv(void* recv, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t s) {
  return p2p(false, nullptr, recv, count, peer, comm, s);
}
ncclResult_t ncclGroupEnd(void) {
  t_in_group = false;
  if (!t_group_comm) return ncclSuccess;
  CollArg a;
  a.type = 2;
  a.p2p = t_p2p;
  return enqueue_coll(t_group_comm, t_group_stream, a);
}

// ------------------------------------------------------------------------------------------------------ test control
void mock_reset(int policy, uint64_t seed, int devices) {
  std::lock_guard<std::mutex> lk(M);
  g_policy = policy;
  g_rng.seed(seed);
  g_devices = devices;
  g_errors = 0;
  g_executed = 0;
  g_h2d_bytes = 0;
}
uint64_t mock_h2d_bytes(void) { return g_h2d_bytes; }
void mock_costs(double per_row, double per_float_moved, double per_float_copied) {
  std::lock_guard<std::mutex> lk(M);
  g_cost_row = per_row, g_cost_float = per_float_moved, g_cost_copy = per_float_copied;
  for (MockStream* st : g_streams) st->t = 0.0;
  g_done_at.clear();
}
double mock_makespan(void) {
  std::lock_guard<std::mutex> lk(M);
  double t = 0.0;
  for (MockStream* st : g_streams) t = std::max(t, st->t);
  return t;
}
int mock_errors(void) { return g_errors; }
uint64_t mock_executed(void) { return g_executed; }

}  // extern "C"

// ------------------------------------------------------------------- the two streaming kernels every build needs
namespace ddt {

hipError_t launch_chain_sum(const float* parts, uint32_t n_parts, size_t n, float* out, bool exact, hipStream_t s, size_t pitch_) {
  Op* op = new Op();
  op->cost = (double)n * n_parts * g_cost_copy;
  const size_t pitch = pitch_ ? pitch_ : n;
  op->run = [=] {
    for (size_t i = 0; i < n; ++i) {
      volatile float acc = parts[i];
      for (uint32_t g = 1; g < n_parts; ++g)  // p0 + p1 + ... (ResultsCombiner.sv:292-311)
        acc = exact ? ref_add_exact(parts[(size_t)g * pitch + i], acc) : acc + parts[(size_t)g * pitch + i];
      out[i] = acc;
    }
  };
  enqueue(s, op);
  return hipSuccess;
}

hipError_t launch_cm_combine(const float* parts0, size_t pitch, size_t n, uint32_t real_groups, uint32_t clusters, bool per_group, bool cm_order, float* out0,
                             bool exact, hipStream_t s, uint32_t n_classes, uint32_t class_positions, size_t out_pitch) {
  Op* op = new Op();
  op->cost = (double)n * (per_group ? real_groups : clusters) * g_cost_copy * n_classes;
  op->run = [=] {
    for (uint32_t k = 0; k < n_classes; ++k)   // (the classes of a one-vs-all model: class_positions partial sums each, class k's sum to out[k][row])
    for (size_t i = 0; i < n; ++i) {
      const float* parts = parts0 + (size_t)k * class_positions * pitch;
      float* out = out0 + (size_t)k * out_pitch;
      volatile float total = 0.0f;
      uint32_t pos = 0;
      for (uint32_t c = 0; c < clusters; ++c) {  // per cluster acc <- p + acc (FPAggregator.v:79-131), then total <- acc + total (Core.sv:486-541)
        const uint32_t len = (real_groups + clusters - 1u - c) / clusters, cnt = per_group ? len : (len ? 1u : 0u);
        if (!cnt) break;
        volatile float acc = 0.0f;
        for (uint32_t j = 0; j < cnt; ++j) {  // the cluster's j-th partial sum: behind one another in a cluster-major image, every C-th group in stream order
          const float p = parts[(size_t)(cm_order ? pos + j : c + j * clusters) * pitch + i];
          acc = exact ? ref_add_exact(p, acc) : p + acc;
        }
        total = exact ? ref_add_exact(acc, total) : acc + total;
        pos += cnt;
      }
      out[i] = total;
    }
  };
  enqueue(s, op);
  return hipSuccess;
}

hipError_t launch_argmax_strided(const float* scores, uint32_t K, size_t pitch, size_t n, int32_t* labels, hipStream_t s) {
  Op* op = new Op();
  op->run = [=] {
    for (size_t i = 0; i < n; ++i) {
      uint32_t best = 0;
      for (uint32_t k = 1; k < K; ++k)
        if (scores[(size_t)k * pitch + i] > scores[(size_t)best * pitch + i]) best = k;  // lowest index wins ties
      labels[i] = (int32_t)best;
    }
  };
  enqueue(s, op);
  return hipSuccess;
}
hipError_t launch_argmax(const float* scores, uint32_t K, size_t n, int32_t* labels, hipStream_t s) { return launch_argmax_strided(scores, K, n, n, labels, s); }

}  // namespace ddt
