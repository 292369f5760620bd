// mock_engine.cpp -- TEST INFRASTRUCTURE (tests/test_comm_mock.py): the deferred-execution HIP / RCCL model plus a STAND-IN
// engine, so that csrc/ddt_comm.cpp can be built and run alone.  A rank's partial score of tuple `row` (word 0 of the tuple
// line) for class k is the integer-valued float f(shard, k, row) below -- sums over ranks are exact in any order, so every
// result is checked bit for bit.
#include "mock_runtime.cpp"

#include <algorithm>

namespace {

// the stand-in engine's arithmetic
inline float partial(uint32_t shard, uint32_t cls, uint32_t row) { return (float)((int)((row * 7u + shard * 13u + cls * 101u) % 1000u) - 500); }
struct MockModel {
  uint32_t shard = 0, count = 1, classes = 1;
};
std::map<ddt_engine*, MockModel> g_models;

}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------- the stand-in engine (C-ABI)
int ddt_create(ddt_engine** out, int device_id) {
  if (!out || device_id < 0 || device_id >= g_devices) return DDT_EINVAL;
  ddt_engine* e = new ddt_engine();
  e->device = device_id;
  *out = e;
  return DDT_OK;
}
void ddt_destroy(ddt_engine* e) {
  if (!e) return;
  {
    std::lock_guard<std::mutex> lk(M);
    g_models.erase(e);
  }
  delete e;
}
static int mock_load(ddt_engine* e, const ddt_params* p, uint32_t classes, uint32_t shard, uint32_t count) {
  if (!e || !p || count == 0 || shard >= count) return DDT_EINVAL;
  e->p = *p;
  e->num_classes = classes;
  e->loaded = true;
  std::lock_guard<std::mutex> lk(M);
  g_models[e] = MockModel{shard, count, classes};
  return DDT_OK;
}
int ddt_load_model_shard(ddt_engine* e, const ddt_params* p, const void*, size_t, const void*, size_t, uint32_t shard, uint32_t count) {
  return mock_load(e, p, 1, shard, count);
}
int ddt_load_model_sparse(ddt_engine* e, const ddt_params* p, const void*, size_t, const uint64_t*, uint32_t shard, uint32_t count) {
  return mock_load(e, p, 1, shard, count);
}
int ddt_load_model_multiclass(ddt_engine* e, const ddt_params* p, const void*, size_t, const void*, size_t, uint32_t classes, int, uint32_t shard,
                              uint32_t count) {
  return mock_load(e, p, classes, shard, count);
}
int ddt_load_model(ddt_engine* e, const ddt_params* p, const void*, size_t, const void*, size_t) { return mock_load(e, p, 1, 0, 1); }
int ddt_get_info(const ddt_engine* e, ddt_info* out) {
  if (!e || !out) return DDT_EINVAL;
  memset(out, 0, sizeof(*out));
  std::lock_guard<std::mutex> lk(M);
  const MockModel m = g_models[const_cast<ddt_engine*>(e)];
  const uint32_t T = e->p.num_trees, per = (T + m.count - 1) / m.count;
  out->tree_begin = std::min(m.shard * per, T);
  out->tree_end = std::min(out->tree_begin + per, T);
  return DDT_OK;
}
int ddt_score(ddt_engine* e, const void* tuple_lines, size_t n, float* scores_out) {  // host buffers, synchronous
  if (!e || !e->loaded) return DDT_ESTATE;
  MockModel m;
  {
    std::lock_guard<std::mutex> lk(M);
    m = g_models[e];
  }
  const uint32_t W = (e->p.num_features + 3u) / 4u * 4u;
  const uint32_t* t = reinterpret_cast<const uint32_t*>(tuple_lines);
  for (size_t i = 0; i < n; ++i) scores_out[i] = partial(m.shard, 0, t[i * W]);
  return DDT_OK;
}
const char* ddt_strerror(int) { return "mock"; }
const char* ddt_last_error(const ddt_engine* e) { return e ? e->err : ""; }

float mock_partial(uint32_t shard, uint32_t cls, uint32_t row) { return partial(shard, cls, row); }

}  // extern "C"

namespace ddt {

uint32_t tuple_words(const ddt_params& p) { return (p.num_features + 3u) / 4u * 4u; }
void engine_enter_collective_job(ddt_engine* e) {
  if (e) e->collective_job = true;
}

int engine_score_device(ddt_engine* e, const void* d_tuples, size_t n, float* d_scores, hipStream_t s) {
  MockModel m;
  {
    std::lock_guard<std::mutex> lk(M);
    m = g_models[e];
  }
  const uint32_t W = tuple_words(e->p);
  const uint32_t* t = reinterpret_cast<const uint32_t*>(d_tuples);
  Op* op = new Op();
  op->cost = (double)n * g_cost_row;
  op->run = [=] {
    for (size_t i = 0; i < n; ++i) d_scores[i] = partial(m.shard, 0, t[i * W]);
  };
  enqueue(s, op);
  return DDT_OK;
}

int engine_classify_device(ddt_engine* e, const void* d_tuples, size_t n, float* d_class_scores, int32_t* d_labels, hipStream_t s) {
  MockModel m;
  {
    std::lock_guard<std::mutex> lk(M);
    m = g_models[e];
  }
  const uint32_t W = tuple_words(e->p);
  const uint32_t* t = reinterpret_cast<const uint32_t*>(d_tuples);
  Op* op = new Op();
  op->cost = (double)n * g_cost_row * m.classes;
  op->run = [=] {
    for (uint32_t k = 0; k < m.classes; ++k)
      for (size_t i = 0; i < n; ++i) d_class_scores[(size_t)k * n + i] = partial(m.shard, k, t[i * W]);
    if (d_labels)
      for (size_t i = 0; i < n; ++i) {
        uint32_t best = 0;
        for (uint32_t k = 1; k < m.classes; ++k)
          if (d_class_scores[(size_t)k * n + i] > d_class_scores[(size_t)best * n + i]) best = k;
        d_labels[i] = (int32_t)best;
      }
  };
  enqueue(s, op);
  return DDT_OK;
}

}  // namespace ddt
