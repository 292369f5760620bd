// ddt_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4).  No MFMA: the path is
// compare + gather.  Measured binding resources on the headline shape (profiles/): the CU's
// LDS pipe (2 DS ops per node visit) together with VALU issue (4 ops per visit, ~4 cycles each per wave and SIMD);
// the rank-quantised depth-8 walk runs at 95 % of its VALU bound (tools/ubench).
//
// Hot path replaced: the DTPU traversal loop + leaf reduce of the reference
//   rtl/DTEngine/core/DTPU.sv:579-760       read node -> gather feature -> compare -> next node -> leaf
//   rtl/DTEngine/core/FPAddersReduceTree.sv:94-141, FPAggregator.v:79-131, Core.sv:486-541   leaf sum
//
// One mapping everywhere (lane = tuple, the features of a tile of tuples in LDS feature-major so that the
// per-lane feature gather is conflict-free for any feature index); the kernels, in file order:
//   score_tile_kernel          fp32 features; the model streams through LDS in double-buffered chunks (global->LDS
//                              DMA), U trees walked concurrently per lane, level loop fully unrolled.
//   score_stream_kernel        small ensembles (whole model resident in LDS): persistent blocks, coalesced tuple
//                              loads prefetched one tile ahead -- the HBM-bound regime.
//   score_q16_kernel, score_q16p_kernel (the rank pre-pass: ddt_prepass.hip)
//                              the rank-quantised path: features replaced exactly by u16 ranks among the model's
//                              thresholds (two pre-pass flavours), 4-byte node records, 32 waves per CU -- the
//                              default for big ensembles.
//   score_generic_kernel       any depth / any feature count; deep trees: top levels staged in LDS, rest gathered.
//   chain_sum_kernel, argmax_kernel, synth_tuples_kernel   multi-device combine, class labels, bench inputs.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "ddt_device.h"
#include "ddt_internal.h"

namespace ddt {

// ---------------------------------------------------------------------------------------------------
// the tile kernel (model streamed through LDS)
// ---------------------------------------------------------------------------------------------------
template <int D, int U, int R, int CT, int BUF_OFF, int PHASE0, bool SLOW, int SUM, bool FUSED>
__device__ __forceinline__ void compute_chunk(const uint32_t (&lane_off)[R], const uint32_t miss_key, const uint32_t C,
                                              RefAcc<R>& ra, double (&dacc)[R], const bool exact) {
  constexpr int TREE_BYTES = 12 << D;
  static_assert(CT % U == 0, "sub-group geometry");
#pragma unroll
  for (int sg = 0; sg < CT / U; ++sg) {
    float lf[R][U];
    walk_trees<D, U, R, TREE_BYTES, SLOW, FUSED>((uint32_t)(BUF_OFF + sg * U * TREE_BYTES), lane_off, miss_key, lf);
    fold_leaves<U, R, SUM>(lf, (PHASE0 + sg) & 1, C, ra, dacc, exact);
  }
}

template <int D, int THREADS, int R, int CT, int U, int STAGE, int OPT>
__global__ __launch_bounds__(THREADS) void score_tile_kernel(const ScoreArgs a) {
  constexpr int TILE = THREADS * R;
  constexpr int TREE_BYTES = 12 << D;
  constexpr int CHUNK_BYTES = TREE_BYTES * CT;
  constexpr int ROW = TILE * 4;
  constexpr bool FUSED = (OPT & 1) != 0;
  constexpr int MB = FUSED ? (4 << D) : 0;  // model buffers start here (Variant::model_base)
  constexpr int FEAT_OFF = (MB + 2 * CHUNK_BYTES + ROW - 1) / ROW * ROW;
  static_assert((ROW & (ROW - 1)) == 0, "tile must be a power of two (row|lane OR trick)");
  static_assert(CT == 4 || CT % 8 == 0, "chunk = half a PU group or whole groups");
  static_assert(CT != 4 || U == 4, "CT=4 needs U=4");
  // dynamic LDS only (launch-time size); addressed absolutely through lds_*()

  const int tid = threadIdx.x;
  const uint64_t tile0 = (uint64_t)blockIdx.x * TILE;
  const uint32_t n_chunks = a.n_chunks;

  StageRegs<THREADS, CHUNK_BYTES> sr;
  if (STAGE == 1) dma_chunk<THREADS, CHUNK_BYTES>(a.img, 0, MB, tid);
  else sr.load(a.img, 0, tid);

  // ---- stage the tuple tile, transposed to [feature][tuple]; detect missing values on the way ----
  const uint32_t W = a.tuple_words;
  uint32_t lane_off[R];
  uint32_t miss_any = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t col = (uint32_t)(r * THREADS + tid);
    lane_off[r] = col * 4u;
    const uint64_t row = tile0 + col;
    const bool valid = row < a.n;
    // Quad-coalesced loads: the four lanes of a quad read 64 contiguous bytes (lines 4g..4g+3) of ONE row per
    // instruction, rows quad_base+0..3 over four instructions, and a 4x4 transpose inside the quad (DPP) hands
    // every lane the four lines of its own row.  One-row-per-lane loads touch 64 cache lines per instruction
    // (8192 L1 accesses per 1024x32 tile, ~7.8 us per tile exposed: profiles/archive/r01_tile_overhead.md); this is 16.
    const uint32_t t4 = (uint32_t)tid & 3u;
    const uint64_t quad_row = tile0 + (uint64_t)(col & ~3u);
    const uint32_t lpt = W / 4u;
    for (uint32_t g0 = 0; g0 < lpt; g0 += 8) {  // two groups of four lines = 8 independent 16-byte loads in flight
      u32x4 v[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t line = g0 + 4u * h + t4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint64_t rj = quad_row + (uint64_t)j;
          v[h][j] = (rj < a.n && line < lpt) ? *reinterpret_cast<const u32x4*>(a.tuples + rj * W + 4u * line)
                                             : u32x4{0u, 0u, 0u, 0u};
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        quad_transpose(v[h], t4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t line = g0 + 4u * h + (uint32_t)i;
          if (line < lpt) {
            const uint32_t fa = (uint32_t)FEAT_OFF + (4u * line) * (uint32_t)ROW + col * 4u;
            lds_st_u32(fa + 0 * ROW, stage_word(v[h][i].x, a, miss_any, valid));
            lds_st_u32(fa + 1 * ROW, stage_word(v[h][i].y, a, miss_any, valid));
            lds_st_u32(fa + 2 * ROW, stage_word(v[h][i].z, a, miss_any, valid));
            lds_st_u32(fa + 3 * ROW, stage_word(v[h][i].w, a, miss_any, valid));
          }
        }
      }
    }
  }
  if (STAGE == 0) sr.commit(MB, tid);
  const bool slow = block_any<THREADS>(miss_any, (uint32_t)FEAT_OFF + W * (uint32_t)ROW, tid);

  RefAcc<R> ra;
  ra.init();
  double dacc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) dacc[r] = 0.0;
  const uint32_t C = a.clusters, miss_key = a.miss_key;
  const int SUM1 = (int)a.sum_mode;  // 0 reference order / IEEE adds, 1 fp64, 2 reference order / reference adder
  const bool exact = SUM1 == 2;

  // chunk k lives in buffer k&1; the loop is unrolled by two so buffer offsets are immediates
#define DDT_COMPUTE(BUF, PH)                                                                                          \
  do {                                                                                                                \
    if (SUM1 != 1) {                                                                                                  \
      if (!slow) compute_chunk<D, U, R, CT, MB + (BUF) * CHUNK_BYTES, PH, false, 0, FUSED>(lane_off, miss_key, C, ra, dacc, exact); \
      else compute_chunk<D, U, R, CT, MB + (BUF) * CHUNK_BYTES, PH, true, 0, FUSED>(lane_off, miss_key, C, ra, dacc, exact);        \
    } else {                                                                                                          \
      if (!slow) compute_chunk<D, U, R, CT, MB + (BUF) * CHUNK_BYTES, PH, false, 1, FUSED>(lane_off, miss_key, C, ra, dacc, false); \
      else compute_chunk<D, U, R, CT, MB + (BUF) * CHUNK_BYTES, PH, true, 1, FUSED>(lane_off, miss_key, C, ra, dacc, false);        \
    }                                                                                                                 \
  } while (0)

  constexpr int PH1 = (CT == 4) ? 1 : 0;  // CT=4: odd chunks are the second half of a PU group
  for (uint32_t k = 0; k < n_chunks; k += 2) {
    if (STAGE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // chunk k is in buffer 0 for everyone; everyone is done with buffer 1
    const bool more1 = k + 1 < n_chunks;
    if (more1) {
      if (STAGE == 1) dma_chunk<THREADS, CHUNK_BYTES>(a.img, k + 1, MB + CHUNK_BYTES, tid);
      else sr.load(a.img, k + 1, tid);
    }
    DDT_COMPUTE(0, 0);
    if (!more1) break;
    if (STAGE == 0) sr.commit(MB + CHUNK_BYTES, tid);
    if (STAGE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool more2 = k + 2 < n_chunks;
    if (more2) {
      if (STAGE == 1) dma_chunk<THREADS, CHUNK_BYTES>(a.img, k + 2, MB, tid);
      else sr.load(a.img, k + 2, tid);
    }
    DDT_COMPUTE(1, PH1);
    if (STAGE == 0 && more2) sr.commit(MB, tid);
  }
#undef DDT_COMPUTE

  ra.align(C);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t row = tile0 + (uint64_t)(r * THREADS + tid);
    if (row < a.n) a.out[row] = (SUM1 != 1) ? ra.total(r, C, exact) : (float)dacc[r];
  }
}

template <int D, int THREADS, int R, int CT, int U, int STAGE, int OPT>
static hipError_t launch_tile(const ScoreArgs& a, const Variant& v, hipStream_t s) {
  auto kern = score_tile_kernel<D, THREADS, R, CT, U, STAGE, OPT>;
  const uint32_t lds = v.lds_bytes(a.tuple_words);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const uint64_t tile = (uint64_t)THREADS * R;
  const uint64_t blocks = (a.n + tile - 1) / tile;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(THREADS), lds, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// the stream kernel: small ensembles whose whole image fits in LDS next to one tile of tuples -- the
// HBM-bound regime (e.g. BASELINE config 1: 8 trees x depth 4 = 32 node visits per 68 compulsory bytes).
// Persistent blocks (grid = CUs x resident blocks per CU) walk the tiles with a grid stride.  Tuples are read with
// fully coalesced, nontemporal 16-byte loads (thread t takes float4 #t of the tile's contiguous byte range), one tile
// AHEAD of the compute, and written transposed into LDS; the rows of tuple line q start 64 q bytes late
// (Variant::feat_word_stream), which makes those stores conflict-free.  Model image: classic layout at LDS [0, img_bytes).
// What bounds it (profiles/EXPERIMENTS.md, round 3): the memory pattern itself.  tools/ubench `tilepat` -- this kernel's loads
// and stores without any compute -- reads 6.6-6.9 TB/s with no result store and 4.7-5.3 TB/s with the 4-byte result per
// 64-byte tuple (any store shape: 4 B or 16 B per lane, nontemporal, 4 KiB bursts); this kernel runs at 4.8-5.3.  Getting
// there took a VALU trim (488 -> 258 VALU instructions per tuple: 2.90 -> 2.73 ms per 200 M tuples).  Not kept, all equal
// within box noise once the trim was in: 7 / 8 waves per SIMD, two tiles in flight per block, quad loads + DPP transpose.
// Round 4, PHASED RESULT STORES: the store's cost is paid in the DRAM, not on the chip (tools/ubench/storephase.hip: the same
// stores into an L2-resident window cost nothing; a vector store, an L2 atomic and a scalar store cost the same; L2 <-> fabric
// write queues idle, read latency unchanged) -- the HBM serves a read stream with a few writes sprinkled in at ~70 % of its
// read-only rate whatever the write share.  So the writes are taken out of the read stream IN TIME: a wave parks its 64 scores
// per tile in LDS (ScoreArgs::stream_res_tiles slots of 256 bytes, private to the wave: no barrier) and all waves of the chip
// write them, nontemporally, when the 100 MHz s_memrealtime clock passes a multiple of ScoreArgs::stream_window_ticks (or the
// wave's slots are full): the memory sees read-only traffic for ~30 us, then a short burst of writes.  Pattern alone: 2.08-2.25 ms
// per 12.8 GB against 2.24-2.56 (direct nontemporal stores) and 2.70 (direct plain stores), box by box.
// ---------------------------------------------------------------------------------------------------
constexpr int kStreamThreads = 256;
constexpr uint32_t kStreamRow = kStreamThreads * 4u;
constexpr uint32_t kStreamResMin = 4, kStreamResMax = 24;  // phased result stores: tiles a wave may park in LDS
constexpr uint32_t kStreamResWant = 10, kStreamFewVisits = 64;  // ... models of at most that many node visits per tuple give up resident blocks (down to 4) for that many slots
constexpr uint32_t kStreamWindowTicks = 3000;              // ... and the write window's period in 10 ns ticks (30 us)

// FIXED: the tuple has exactly MAXLPT lines (config 1: 16 features = 4 lines), so element -> (tuple, line) is a shift and the
// sixteen LDS stores of a thread share one address register.  Round 3 counters (profiles/archive/r03_pmc_stream_cfg1.md) showed this
// kernel VALU-bound, not HBM-bound: 488 VALU instructions per tuple = 92 % VALU issue; staging (runtime division, per-word
// missing bookkeeping) and the accumulator ring's align() were 60 % of them.
template <int D, int U, int MAXLPT, bool FIXED>
__device__ __forceinline__ void stream_body(const ScoreArgs& a) {
  constexpr int THREADS = kStreamThreads, TILE = kStreamThreads;
  constexpr int TREE_BYTES = 12 << D;
  const int tid = threadIdx.x;
  const uint32_t W = a.tuple_words, LPT = FIXED ? (uint32_t)MAXLPT : W / 4u;
  const uint32_t img_bytes = a.n_trees * (uint32_t)TREE_BYTES;
  const uint32_t feat_off = (img_bytes + kStreamRow - 1u) / kStreamRow * kStreamRow;  // == host's Variant::feat_off
  const uint64_t n_tiles = (a.n + TILE - 1) / TILE;
  const uint32_t C = a.clusters, miss_key = a.miss_key, miss_raw = a.miss_raw;

  // resident model
  for (uint32_t off = (uint32_t)tid * 16u; off < img_bytes; off += THREADS * 16u)
    lds_st_u4(off, a.img[off / 16u]);
  __syncthreads();  // the only block barrier: from here on the waves run on their own

  // element e (0..TILE*LPT) of a tile = float4 #e of its byte range: tuple e / LPT, line e % LPT.  WAVE-PRIVATE tiles (round 4): a
  // wave loads, stages and walks the SAME 64 tuples -- lane l takes elements 64 w LPT + l + 64 i of the tile, the contiguous
  // 64 LPT float4 of the wave's tuples (coalesced as before) -- so nothing a wave reads from the tile was written by another wave
  // and the two block barriers per tile (tile consumed / tile staged + the block-wide missing flag) are gone: the missing flag
  // is a wave ballot, the waves of a block drift apart and fill each other's stalls.
  const uint32_t wave_e0 = (uint32_t)(tid & ~63) * LPT + (uint32_t)(tid & 63);
  auto prefetch = [&](uint64_t tile, u32x4 (&pre)[MAXLPT]) {
    // nontemporal: the tuple stream is read once (tools/ubench `hbm`: 6.5 vs 5.9 TB/s read-only; this kernel on config 1: 2.90 vs 3.04 ms)
    const u32x4* src = reinterpret_cast<const u32x4*>(a.tuples + tile * TILE * W);
    const uint64_t rows_left = a.n - tile * TILE;
    if (rows_left >= (uint64_t)TILE) {  // wave-uniform: a full tile loads unguarded
#pragma unroll
      for (int i = 0; i < MAXLPT; ++i)
        if (FIXED || (uint32_t)i < LPT) pre[i] = __builtin_nontemporal_load(src + (wave_e0 + (uint32_t)i * 64u));
    } else {
      const uint32_t avail = (uint32_t)rows_left * LPT;
#pragma unroll
      for (int i = 0; i < MAXLPT; ++i) {
        const uint32_t e = wave_e0 + (uint32_t)i * 64u;
        if ((FIXED || (uint32_t)i < LPT) && e < avail) pre[i] = __builtin_nontemporal_load(src + e);
        else pre[i] = u32x4{0u, 0u, 0u, 0u};  // rows past n: zeros (may only make the last tile take the slow walk)
      }
    }
  };
  // raw / IEEE-key transform of the staged words + missing detection; the detection is a compare into a lane mask that
  // the scalar unit ORs together
  auto stage = [&](auto ieee_tag, const u32x4 (&pre)[MAXLPT]) -> bool {
    constexpr bool IEEE = decltype(ieee_tag)::value;
    bool miss_any = false;
#pragma unroll
    for (int i = 0; i < MAXLPT; ++i) {
      if (FIXED || (uint32_t)i < LPT) {
        const uint32_t e = wave_e0 + (uint32_t)i * 64u;
        const uint32_t t = e / LPT, q = e - t * LPT;
        const uint32_t fa = feat_off + (4u * q) * kStreamRow + q * kStreamSkew + t * 4u;  // Variant::feat_word_stream
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          uint32_t v = pre[i][c];
          const bool m = v == miss_raw;
          miss_any |= m;
          if (IEEE) v = m ? kMissSentinelIeee : ieee_key(v);
          lds_st_u32(fa + (uint32_t)c * kStreamRow, v);
        }
      }
    }
    return miss_any;
  };
  const uint64_t G = gridDim.x;
  // phased result stores (see above): NB slots of one tile's 64 scores per wave; 32-bit clock arithmetic (wrap-safe differences)
  const uint32_t NB = a.stream_res_tiles, window = a.stream_window_ticks;
  auto ring_addr = [&]() -> uint32_t {  // (recomputed at a tile's end: one VGPR less across the walk; the asm keeps hipcc from hoisting it)
    uint32_t t = (uint32_t)tid;
    asm volatile("" : "+v"(t));
    return a.stream_res_off + (t >> 6) * NB * 256u + (t & 63u) * 4u;
  };
  uint32_t r_count = 0, deadline = 0;
  uint64_t r_first = blockIdx.x;
  if (NB) {
    // windows are aligned to multiples of `window` in ABSOLUTE time, chip-wide: the phase comes from the 64-bit clock (the low 32 bits wrap
    // every 43 s, and a window that does not divide 2^32 would shift its phase there); differences stay 32-bit
    const uint64_t now64 = __builtin_amdgcn_s_memrealtime();
    const uint32_t now = (uint32_t)now64;
    deadline = now - (uint32_t)(now64 % (uint64_t)window) + window;
  }
  auto flush = [&](uint64_t next_first) {
    const uint32_t ring = ring_addr();
    for (uint32_t j = 0; j < r_count; ++j) {
      const uint64_t row = (r_first + (uint64_t)j * G) * TILE + (uint64_t)tid;
      if (row < a.n) __builtin_nontemporal_store(lds_f32(ring + j * 256u), a.out + row);
    }
    r_count = 0;
    r_first = next_first;
  };
  // one tile: stage `pre`, refill it with the block's next tile (its HBM reads fly during the walks), walk, store
  auto step = [&](uint64_t tile, u32x4 (&pre)[MAXLPT]) {
    const bool miss_any = a.ieee ? stage(std::true_type{}, pre) : stage(std::false_type{}, pre);
    const bool slow = __ballot(miss_any) != 0ull;  // wave-uniform: one of the wave's 64 tuples holds a missing value
    // the walk reads what OTHER lanes of this wave staged: the LDS executes a wave's instructions in order; the fence keeps hipcc from
    // moving the reads in front of the stores (wavefront scope: no instruction)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (tile + G < n_tiles) prefetch(tile + G, pre);

    RefAcc<1> ra;
    ra.init();
    double dacc[1] = {0.0};
    const uint32_t lane_off[1] = {(uint32_t)tid * 4u};
    for (uint32_t t0 = 0; t0 < a.n_trees; t0 += (uint32_t)U) {  // U = 8: one PU group per iteration, U = 4: half
      float lf[1][U];
      const uint32_t base = t0 * (uint32_t)TREE_BYTES;
      const int phase = (int)((t0 / (uint32_t)U) & 1u);
      if (!slow) walk_trees<D, U, 1, TREE_BYTES, false, false, true>(base, lane_off, miss_key, lf);
      else walk_trees<D, U, 1, TREE_BYTES, true, false, true>(base, lane_off, miss_key, lf);
      if (a.sum_mode != 1) fold_leaves<U, 1, 0>(lf, phase, C, ra, dacc, a.sum_mode == 2);
      else fold_leaves<U, 1, 1>(lf, phase, C, ra, dacc);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next tile's staging stays behind this tile's walk)
    __builtin_amdgcn_wave_barrier();
    const float score = (a.sum_mode != 1) ? ra.total_ring(0, C, a.sum_mode == 2) : (float)dacc[0];
    if (NB == 0u) {
      const uint64_t row = tile * TILE + (uint64_t)tid;
      if (row < a.n) a.out[row] = score;
    } else {
      lds_st_u32(ring_addr() + r_count * 256u, __float_as_uint(score));
      ++r_count;
      const uint32_t now = (uint32_t)__builtin_amdgcn_s_memrealtime();
      const bool due = (int32_t)(now - deadline) >= 0;  // wave-uniform
      if (due) deadline = now - (now - deadline) % window + window;  // the next multiple (the phase of the old deadline is kept)
      if (due || r_count == NB) flush(tile + G);
    }
  };

  uint64_t tile = blockIdx.x;
  u32x4 pre[MAXLPT];
  if (tile < n_tiles) prefetch(tile, pre);
  for (; tile < n_tiles; tile += G) step(tile, pre);
  if (NB) flush(0);
}

template <int D, int U, int MAXLPT>
// (6 blocks per CU only where the walk fits 80 VGPRs: the depth-6 instance spilled two registers under that bound)
__global__ __launch_bounds__(kStreamThreads, (MAXLPT <= 4 && D <= 4) ? 6 : 4) void score_stream_kernel(const ScoreArgs a) {
  if (a.tuple_words == 4u * (uint32_t)MAXLPT) stream_body<D, U, MAXLPT, true>(a);
  else stream_body<D, U, MAXLPT, false>(a);
}

uint32_t stream_blocks_per_cu(uint32_t lds_bytes) {
  uint32_t b = (160u * 1024u) / (lds_bytes ? lds_bytes : 1u);
  return b < 1u ? 1u : (b > 8u ? 8u : b);
}

template <int D, int U, int MAXLPT>
static hipError_t launch_stream(const ScoreArgs& a, const Variant& v, hipStream_t s) {
  if (a.tuple_words > 4u * MAXLPT) return hipErrorInvalidValue;
  auto kern = score_stream_kernel<D, U, MAXLPT>;
  const uint32_t lds = v.lds_bytes_stream(a.n_trees, a.tuple_words);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const uint64_t tiles = (a.n + kStreamThreads - 1) / kStreamThreads;
  if (tiles == 0) return hipSuccess;
  // persistent: CUs x the blocks that are actually resident per CU (registers bound this before the LDS does: a grid of
  // LDS-many blocks per CU would run as one full wave of blocks and a thinner second one)
  // (these caches are per kernel instantiation -- statics of a function template -- and keyed on the device as well: another GPU of the
  // process may hold a different number of blocks)
  int dev = -1;
  (void)hipGetDevice(&dev);
  static thread_local uint32_t occ_lds = ~0u, occ_blocks = 0;
  static thread_local int occ_dev = -2;
  if (occ_lds != lds || occ_dev != dev) {
    occ_dev = dev;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), kStreamThreads, lds) != hipSuccess || occ < 1)
      occ = (int)stream_blocks_per_cu(lds);
    occ_lds = lds;
    occ_blocks = (uint32_t)occ;
  }
  uint32_t per_cu = stream_blocks_per_cu(lds);
  if (a.stream_blocks_per_cu) per_cu = a.stream_blocks_per_cu;  // option "stream_blocks_per_cu" (A/B)
  else if (occ_blocks < per_cu) per_cu = occ_blocks;
  // phased result stores: the LDS the resident blocks leave free holds the waves' score slots (1 KiB per tile and block).
  // option "stream_res_tiles": 0 = as many as fit (up to kStreamResMax; fewer than kStreamResMin: direct stores), 1 = direct
  // stores (A/B), n = at most n.  Only when a block has enough tiles for the windows to matter.
  ScoreArgs b = a;
  b.stream_res_tiles = 0;
  b.stream_res_off = (lds + 255u) & ~255u;
  b.stream_window_ticks = a.stream_window_ticks ? a.stream_window_ticks : kStreamWindowTicks;
  uint32_t lds_launch = lds;
  if (a.stream_res_tiles != 1u) {
    const uint32_t cap = a.stream_res_tiles ? a.stream_res_tiles : kStreamResMax;
    // a block's LDS is allocated in granules (1280 bytes assumed: with 512 the occupancy query said five blocks of 32256 bytes fit
    // a CU, and four ran)
    auto slots_for = [&](uint32_t blocks) -> uint32_t {
      const uint32_t budget = (160u * 1024u) / blocks / 1280u * 1280u;
      const uint32_t nb = budget > b.stream_res_off ? (budget - b.stream_res_off) / 1024u : 0u;
      return nb > cap ? cap : nb;
    };
    uint32_t pc = per_cu;
    // few node visits per tuple (BASELINE config 1's regime, HBM-bound): one resident block fewer buys more slots than it costs --
    // 8 trees x depth 4 x 16 features, 200 M tuples on one box: 6 blocks x 7 slots 2.58-2.65 ms, 5 x 12 2.41-2.44, 4 x 21 2.45-2.52,
    // direct stores 2.71-2.90 (profiles/r04_stream_phased_stores.md)
    if (!a.stream_blocks_per_cu && a.n_trees * a.levels <= kStreamFewVisits)
      while (pc > 4u && slots_for(pc) < kStreamResWant) --pc;
    uint32_t nb = slots_for(pc);
    if (nb >= kStreamResMin && tiles >= 8ull * a.num_cus * pc) {
      static thread_local uint32_t ok_lds = ~0u, ok_pc = 0, ok_cap = 0, ok_nb = 0;  // the occupancy query per (device, lds, blocks, cap)
      static thread_local int ok_dev = -2;
      if (ok_lds != lds || ok_pc != pc || ok_cap != cap || ok_dev != dev) {
        ok_dev = dev;
        for (; nb >= kStreamResMin; --nb) {  // the blocks the grid counts on must be resident together
          int occ = 0;
          const uint32_t want = b.stream_res_off + nb * 1024u;
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) != hipSuccess) continue;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), kStreamThreads, want) == hipSuccess && (uint32_t)occ >= pc) break;
        }
        ok_lds = lds, ok_pc = pc, ok_cap = cap, ok_nb = nb >= kStreamResMin ? nb : 0u;
      }
      if (ok_nb) {
        per_cu = pc;
        b.stream_res_tiles = ok_nb;
        lds_launch = b.stream_res_off + ok_nb * 1024u;
      }
    }
  }
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch);
  if (e != hipSuccess) return e;
  uint64_t grid = (uint64_t)a.num_cus * per_cu;
  if (grid > tiles) grid = tiles;
  hipLaunchKernelGGL(kern, dim3((uint32_t)grid), dim3(kStreamThreads), lds_launch, s, b);
  return hipGetLastError();
}

// (the rank pre-pass of the rank-quantised path -- transpose_kernel, rank_kernel, fused_rank_kernel, grouped_rank_kernel, launch_q16_prepass --
// lives in ddt_prepass.hip since round 6; the scoring kernels below DMA the u16 tiles it writes)


// "_gl": the leaves stay in the global image and are gathered through a buffer resource over it: leaf m4 / 4 - 2^D of tree u of
// the sub-group whose leaves start `soff + 4 * 2^D` bytes into the image.  m4 is already a byte offset, so the gather is
// buffer_load_dword v, v_m4, s[rsrc], s_soff offen offset:u*4*2^D -- no address arithmetic on the VALU at all (the pointer form
// cost 24 of a PU group's 303 VALU instructions: a 64-bit add per gather plus the chunk offset)
struct LeafSrc {
  __amdgpu_buffer_rsrc_t rsrc;  // the whole image
  uint32_t soff;                // wave-uniform: byte offset of the sub-group's leaves, minus 4 * 2^D
};
template <int D>
__device__ __forceinline__ float gather_leaf(const LeafSrc& g, int u, uint32_t m4) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g.rsrc, m4 + (uint32_t)(u * (4 << D)), g.soff, 0));
}

// walk over 4-byte records {R (lo16), feature row byte offset (hi16, bit 16 = miss_right in the slow image)};
// m4 = 4 * (1-based heap index); leaves start at byte 4*2^D of the tree, so leaf address = tree + m4.
// WIDE ("q16w_*", Variant::opt bit 6): tuples of 33..64 words -- the row offset j * 2048 no longer fits the record's 16-bit field, so the
// record carries HALF of it (j * 1024; bit 0 stays free for the slow image's miss_right flag) and the walk shifts it back: one more VALU
// instruction per visit, against the fp32 tile kernels at 8 waves per CU that such tuples ran on before
template <int D, int U, int TREE_BYTES, int FEAT_OFF, bool SLOW, bool GL = false, bool WIDE = false>
__device__ __forceinline__ void walk_trees_q16(const uint32_t base, const uint32_t lane2, float (&leaf)[U], const LeafSrc& gleaf) {
  uint32_t m4[U];
#pragma unroll
  for (int u = 0; u < U; ++u) m4[u] = 4u;
#pragma unroll
  for (int lvl = 0; lvl < D; ++lvl) {
    uint32_t nd[U], f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) nd[u] = lds_u32(m4[u] + (base + (uint32_t)(u * TREE_BYTES)));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t off = SLOW ? ((nd[u] >> 16) & 0xFFFEu) : (nd[u] >> 16);
      if (WIDE) off <<= 1;
      // ds_read_u16 with FEAT_OFF as the DS immediate; conflict-free (bank = lane/2, two lanes share a dword)
      f[u] = *reinterpret_cast<const DDT_LDS(uint16_t)*>((off | lane2) + (uint32_t)FEAT_OFF);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool right = f[u] >= (nd[u] & 0xFFFFu);
      if (SLOW) right = (f[u] == kQMissing) ? ((nd[u] >> 16) & 1u) != 0u : right;
      m4[u] = (m4[u] << 1) + (right ? 4u : 0u);
    }
  }
  if (GL) {  // leaves of this sub-group in global memory, 2^D floats per tree: one 4-byte gather per tree on the vector-memory path
#pragma unroll
    for (int u = 0; u < U; ++u) leaf[u] = gather_leaf<D>(gleaf, u, m4[u]);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) leaf[u] = lds_f32(m4[u] + (base + (uint32_t)(u * TREE_BYTES)));
  }
}

// "_s2": the records of levels 0 and 1 are wave-uniform data, so they do not have to come out of LDS: one s_load_dwordx4 per
// tree fetches {padding, root, left, right} (bytes 0..15 of the tree's records in the GLOBAL image) through the scalar cache
// into SGPRs, one sub-group of U trees ahead of the walk.  6 node reads per depth-8 tree on the LDS pipe instead of 8, for
// two v_mov + one v_cndmask more on the VALU (the level-1 record is selected per lane).  The loads go through inline asm:
// hipcc neither sinks them to their first use nor turns every LDS wait into lgkmcnt(0) while they are in flight (an older
// outstanding SMEM only makes hipcc's own counted LDS waits stricter, never unsafe; top_wait() is the one full wait).
// tools/ubench `walk`: 8.17 vs 7.84 T node visits/s for the bare walk.
template <int U>
struct TopRecs {
  u32x4 t[U];
};
template <int TREE_STRIDE>
__device__ __forceinline__ void top_issue(TopRecs<4>& q, const void* tree0) {
  asm volatile("s_load_dwordx4 %0, %4, 0x0\n\ts_load_dwordx4 %1, %4, %5\n\ts_load_dwordx4 %2, %4, %6\n\ts_load_dwordx4 %3, %4, %7"
               : "=&s"(q.t[0]), "=&s"(q.t[1]), "=&s"(q.t[2]), "=&s"(q.t[3])
               : "s"(tree0), "i"(TREE_STRIDE), "i"(2 * TREE_STRIDE), "i"(3 * TREE_STRIDE));
}
__device__ __forceinline__ void top_wait(TopRecs<4>& q) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q.t[0]), "+s"(q.t[1]), "+s"(q.t[2]), "+s"(q.t[3]));
}

template <int D, int U, int TREE_BYTES, int FEAT_OFF, bool SLOW, bool GL, bool WIDE = false>
__device__ __forceinline__ void walk_trees_q16_s2(const TopRecs<U>& q, const uint32_t base, const uint32_t lane2, float (&leaf)[U],
                                                  const LeafSrc& gleaf) {
  static_assert(D >= 3, "two scalar levels + at least one LDS level");
  uint32_t m4[U], nd[U], f[U];
  auto rank_at = [&](uint32_t rec) -> uint32_t {
    uint32_t off = SLOW ? ((rec >> 16) & 0xFFFEu) : (rec >> 16);
    if (WIDE) off <<= 1;
    return *reinterpret_cast<const DDT_LDS(uint16_t)*>((off | lane2) + (uint32_t)FEAT_OFF);
  };
  auto goes_right = [&](uint32_t rec, uint32_t fv) -> bool {
    bool right = fv >= (rec & 0xFFFFu);
    if (SLOW) right = (fv == kQMissing) ? ((rec >> 16) & 1u) != 0u : right;
    return right;
  };
#pragma unroll
  for (int u = 0; u < U; ++u) f[u] = rank_at(q.t[u].y);  // level 0: the root, uniform
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool r0 = goes_right(q.t[u].y, f[u]);
    nd[u] = r0 ? q.t[u].w : q.t[u].z;  // level-1 record
    m4[u] = r0 ? 12u : 8u;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) f[u] = rank_at(nd[u]);
#pragma unroll
  for (int u = 0; u < U; ++u) m4[u] = (m4[u] << 1) + (goes_right(nd[u], f[u]) ? 4u : 0u);
#pragma unroll
  for (int lvl = 2; lvl < D; ++lvl) {
#pragma unroll
    for (int u = 0; u < U; ++u) nd[u] = lds_u32(m4[u] + (base + (uint32_t)(u * TREE_BYTES)));
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = rank_at(nd[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) m4[u] = (m4[u] << 1) + (goes_right(nd[u], f[u]) ? 4u : 0u);
  }
  if (GL) {
#pragma unroll
    for (int u = 0; u < U; ++u) leaf[u] = gather_leaf<D>(gleaf, u, m4[u]);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) leaf[u] = lds_f32(m4[u] + (base + (uint32_t)(u * TREE_BYTES)));
  }
}

// The same walk with the ORDER OF THE LDS READS pinned (round 4).  Left to itself hipcc interleaves only two of the four trees of a
// sub-group (26 VGPRs: counted waits lgkmcnt(1)), and none at all once a kernel carries more live registers (the persistent kernel
// with its 16 prefetch registers: every wait lgkmcnt(0), one chain at a time, 9 % slower).  Here every stage issues its four reads
// back to back -- 4 x node record, 4 x feature rank, ... -- and a sched_barrier that only VALU / SALU / VMEM instructions may
// cross keeps the stages apart: a wave has four dependent chains in flight, the waits come out as lgkmcnt(3).
#define DDT_PIN_DS() __builtin_amdgcn_sched_barrier(0x0016)
template <int D, int U, int TREE_BYTES, int FEAT_OFF, bool SLOW, bool GL, bool WIDE = false>
__device__ __forceinline__ void walk_trees_q16_s2_pin(const TopRecs<U>& q, const uint32_t base, const uint32_t lane2, float (&leaf)[U],
                                                      const LeafSrc& gleaf) {
  static_assert(D >= 3, "two scalar levels + at least one LDS level");
  uint32_t m4[U], nd[U], f[U];
  auto rank_at = [&](uint32_t rec) -> uint32_t {
    uint32_t off = SLOW ? ((rec >> 16) & 0xFFFEu) : (rec >> 16);
    if (WIDE) off <<= 1;
    return *reinterpret_cast<const DDT_LDS(uint16_t)*>((off | lane2) + (uint32_t)FEAT_OFF);
  };
  auto goes_right = [&](uint32_t rec, uint32_t fv) -> bool {
    bool right = fv >= (rec & 0xFFFFu);
    if (SLOW) right = (fv == kQMissing) ? ((rec >> 16) & 1u) != 0u : right;
    return right;
  };
#pragma unroll
  for (int u = 0; u < U; ++u) f[u] = rank_at(q.t[u].y);  // level 0: the root, uniform
  DDT_PIN_DS();
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool r0 = goes_right(q.t[u].y, f[u]);
    nd[u] = r0 ? q.t[u].w : q.t[u].z;  // level-1 record
    m4[u] = r0 ? 12u : 8u;
    f[u] = rank_at(nd[u]);
  }
  DDT_PIN_DS();
#pragma unroll
  for (int u = 0; u < U; ++u) {
    m4[u] = (m4[u] << 1) + (goes_right(nd[u], f[u]) ? 4u : 0u);
    nd[u] = lds_u32(m4[u] + (base + (uint32_t)(u * TREE_BYTES)));
  }
  DDT_PIN_DS();
#pragma unroll
  for (int lvl = 2; lvl < D; ++lvl) {
#pragma unroll
    for (int u = 0; u < U; ++u) f[u] = rank_at(nd[u]);
    DDT_PIN_DS();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      m4[u] = (m4[u] << 1) + (goes_right(nd[u], f[u]) ? 4u : 0u);
      if (lvl + 1 < D) nd[u] = lds_u32(m4[u] + (base + (uint32_t)(u * TREE_BYTES)));
      else leaf[u] = GL ? gather_leaf<D>(gleaf, u, m4[u]) : lds_f32(m4[u] + (base + (uint32_t)(u * TREE_BYTES)));
    }
    DDT_PIN_DS();
  }
}

template <int D, int CT, int U, int OPT = 0>
__global__ __launch_bounds__(kQTile, 8) void score_q16_kernel(const ScoreArgs a, const Q16Aux x) {  // 8 waves per SIMD = two blocks per CU
  constexpr int THREADS = kQTile;
  constexpr bool GL = (OPT & 1) != 0, S2 = (OPT & 2) != 0, HOTDISP = S2, CM = (OPT & 4) != 0, PIN = (OPT & 16) != 0, WIDE = (OPT & 64) != 0;
  // SPLIT (round 6, small batches): blockIdx.y = a SLICE of the cluster-major image.  A batch of a few tiles leaves most of the chip idle
  // while one block walks the whole ensemble (0.35 ms per call at 1000 trees).  The walks are independent; only the adds have an order
  // (FPAggregator.v:79-131: per cluster acc <- x_g + acc over its PU groups g; Core.sv:486-541: total <- acc_c + total over the clusters),
  // so the adds are what is left to a second kernel (launch_cm_combine) that runs them in exactly that order:
  //   Q16Aux::split_len == 0  a slice = a CLUSTER's run of PU groups (= chunks); the block carries the cluster's accumulator as the uncut
  //                           launch would and leaves acc_c in out[c][row]: one partial sum per cluster
  //   Q16Aux::split_len  > 0  a slice = split_len consecutive chunks; the block leaves every group's x_g (its 8-leaf reduce tree, + 0) in
  //                           out[position of the group in the image][row]: up to one block per (tile, PU group) for batches of a tile or two
  constexpr bool SPLIT = (OPT & 128) != 0;
  static_assert(!SPLIT || CT % 8 == 0, "a slice = whole chunks = whole PU groups");  // (image in stream order, !CM: a partial sum per group only)
  static_assert(!S2 || U == 4, "_s2: 4 trees in flight");
  static_assert(!PIN || S2, "the pinned read order exists for the _s2 walk");
  constexpr int TREE_BYTES = GL ? (4 << D) : (8 << D);  // bytes of a tree in LDS (GL: node records only)
  constexpr int CHUNK_BYTES = TREE_BYTES * CT;          // bytes of a chunk in LDS
  constexpr int GCHUNK_UNITS = (8 << D) * CT / 16;      // 16-byte units of a chunk in the global image
  constexpr int FEAT_OFF = 2 * CHUNK_BYTES;
  constexpr int ROW = kQTile * 2;
  static_assert(CT % U == 0 && (U == 4 || U == 8) && (CT == 4 || CT % 8 == 0), "geometry");
  static_assert(FEAT_OFF % ROW == 0, "row|lane OR trick");
  const int tid = threadIdx.x;
  const uint64_t tile = blockIdx.x, tile0 = tile * kQTile;
  const uint32_t W = a.tuple_words;
  uint32_t n_chunks = a.n_chunks;
  constexpr int GSKIP = GCHUNK_UNITS - CHUNK_BYTES / 16;  // dma_chunk strides by the LDS chunk: skip the leaves of the chunks before
  // block-uniform: the tile holds a missing value.  readfirstlane: the flag arrives through a vector load, and without it hipcc
  // treats `img` (and every leaf-gather address derived from it) as lane-varying: 64-bit VALU address arithmetic per gather
  const bool slow = __builtin_amdgcn_readfirstlane((int)x.tile_flags[tile]) != 0;
  const uint4* img = slow ? x.img_slow : a.img;
  uint32_t sp_pos = 0, sp_first = 0;  // SPLIT: which partial sum this block writes next; the slice's first chunk
  const uint64_t sp_row = tile0 + (uint64_t)tid;
  if constexpr (SPLIT) {
    if (x.split_len) {  // split_len consecutive chunks, a partial sum per PU group (at its position in the image)
      const uint32_t real_chunks = (x.real_groups * 8u + (uint32_t)CT - 1u) / (uint32_t)CT;
      sp_first = blockIdx.y * x.split_len;
      n_chunks = real_chunks - sp_first < x.split_len ? real_chunks - sp_first : x.split_len;  // >= 1: the host launches ceil(real chunks / split_len) slices
      sp_pos = sp_first * (uint32_t)(CT / 8);
    } else {            // (cluster-major images only) cluster blockIdx.y's run of PU groups = chunks: cluster j holds (real + C - 1 - j) / C of them (ddt_image.cpp cm_position)
      const uint32_t lg = (uint32_t)__builtin_ctz(a.clusters | 0x100u);
      for (uint32_t j = 0; j < blockIdx.y; ++j) sp_first += (x.real_groups + a.clusters - 1u - j) >> lg;
      n_chunks = (x.real_groups + a.clusters - 1u - blockIdx.y) >> lg;  // >= 1: the host launches min(C, real groups) clusters
      sp_pos = blockIdx.y;
    }
    img += (size_t)sp_first * GCHUNK_UNITS;
  }

  dma_chunk<THREADS, CHUNK_BYTES>(img, 0, 0, tid);
  {  // the whole feature tile is one contiguous block of W*2048 bytes: DMA it in
    const uint4* src = reinterpret_cast<const uint4*>(x.q + tile * (uint64_t)W * kQTile);
    const uint32_t units = W * (ROW / 16);
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    for (uint32_t u0 = 0; u0 < units; u0 += THREADS) {
      const uint32_t lds_addr = (uint32_t)FEAT_OFF + (u0 + (uint32_t)wave_base) * 16u;
      const uint4* g = src + (u0 + (uint32_t)tid);
      if (u0 + (uint32_t)wave_base < units)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds_addr), "v"(g) : "memory");
    }
  }
  RefAcc<1> ra;
  ra.init();
  double dacc[1] = {0.0};
  // "_cm" (cluster-major image): the PU groups of cluster 0 come first in the image, then cluster 1's, ... (each in its slot order,
  // which is all the reference's accumulate needs: FPAggregator.v:79-131 keeps one accumulator per cluster; Core.sv:486-541 adds
  // the clusters in order afterwards).  The kernel then carries ONE accumulator and a running total instead of a ring of C that is
  // rotated after every group (8 v_mov per PU group at C = 8): at a cluster's last group, total <- acc + total, acc <- 0.
  const uint32_t Cc = SPLIT ? 1u : a.clusters, C = CM ? 1u : Cc, lane2 = (((uint32_t)tid & 511u) << 2) | (((uint32_t)tid >> 9) << 1);  // see rank_kernel
  const uint32_t cm_lg = (uint32_t)__builtin_ctz(Cc | 0x100u), cm_real = SPLIT ? (x.split_len ? 1u : n_chunks) : x.real_groups;  // SPLIT: the groups up to the first store
  uint32_t cm_groups = 0, cm_cluster = 0, cm_bound = (cm_real + Cc - 1u) >> cm_lg;  // wave-uniform: groups done, cluster, its end
  float cm_total = 0.f;
  if constexpr (CM) {  // a later part of an ensemble scored in parts (Q16Aux): take up the sum where the launch before left it
    if (x.group0) {
      cm_groups = x.group0;
      while (cm_cluster < Cc && cm_groups >= cm_bound) {  // (a cluster that ended exactly there was closed by that launch)
        ++cm_cluster;
        cm_bound += cm_cluster < Cc ? (cm_real + Cc - 1u - cm_cluster) >> cm_lg : 0u;
      }
    }
    if (x.state_in) {
      const uint64_t r0 = tile0 + (uint64_t)tid;  // (rows of the padding read the workspace's own padding: never stored)
      ra.a[0][0] = x.state_in[r0];
      cm_total = x.state_in[x.n_pad + r0];
    }
  }
  const int SUM1 = (int)a.sum_mode;  // 0 reference order / IEEE adds, 1 fp64, 2 reference order / reference adder
  const bool exact = SUM1 == 2;
  // Sub-groups that hold a real tree (Q16Aux::walk_subgroups): the EMPTY padding behind them is not walked.  A branch per sub-group,
  // so only where the chunk body is no single basic block worth keeping (the depth-7/8 hot path is: +5 %; at depth 6 it measured
  // equal) -- and never as a second copy of the chunk code: at the join of two copies hipcc moves the _s2 record sets while their
  // loads are in flight (tools/check_s2_isa.py refused that build).  Shallow kernels are also the ones with many trees per chunk
  // (16..128), i.e. with the most padding to lose.
  constexpr bool TAILSKIP = D <= 6;
  uint32_t walk_sgs = x.walk_subgroups ? x.walk_subgroups : 0xFFFFFFFFu;
  if constexpr (SPLIT) {
    if (x.walk_subgroups) walk_sgs -= sp_first * (uint32_t)(CT / U);  // (> 0: every chunk of a slice holds a real tree)
  }

  // _s2: two SGPR sets take turns (even / odd sub-group of a chunk; a chunk has an even number of sub-groups, so every chunk
  // starts on top_a): one holds the level-0/1 records of the sub-group being walked, the other receives the next sub-group's
  // (same chunk, or the first of the next chunk; past the end: chunk 0 again, never used), requested before the walk.  No set
  // is ever copied: a copy would read registers whose loads may still be in flight.
  static_assert(!S2 || (CT / U) % 2 == 0, "_s2: even number of sub-groups per chunk");
  TopRecs<4> top_a, top_b;
  // _gl: buffer resource over the image in use (built from wave-uniform values only); non-_gl kernels never load through it
  const __amdgpu_buffer_rsrc_t leaf_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(img), 0, (int)(n_chunks * (uint32_t)(GCHUNK_UNITS * 16)), 0x00020000);
#define DDT_QCOMPUTE(BUF, PH, KIDX)                                                                    \
  do {                                                                                                 \
    _Pragma("unroll") for (int sg = 0; sg < CT / U; ++sg) {                                            \
      float lf[1][U];                                                                                  \
      const LeafSrc gl = {leaf_rsrc, (uint32_t)(KIDX) * (uint32_t)(GCHUNK_UNITS * 16) + (uint32_t)((CT + sg * U - 1) * (4 << D))}; \
      const bool pad_sg = TAILSKIP && (uint32_t)(KIDX) * (uint32_t)(CT / U) + (uint32_t)sg >= walk_sgs; /* wave-uniform */ \
      if constexpr (S2) {                                                                              \
        const uint32_t kn = (sg + 1 < CT / U) ? (uint32_t)(KIDX) : ((uint32_t)(KIDX) + 1u < n_chunks ? (uint32_t)(KIDX) + 1u : 0u); \
        const int sn = (sg + 1 < CT / U) ? sg + 1 : 0;                                                 \
        TopRecs<4>& top_cur = (sg & 1) ? top_b : top_a;                                                \
        TopRecs<4>& top_nxt = (sg & 1) ? top_a : top_b;                                                \
        top_wait(top_cur);                                                                             \
        top_issue<TREE_BYTES>(top_nxt, img + (size_t)kn * GCHUNK_UNITS + (size_t)(sn * U) * (TREE_BYTES / 16)); \
        if (pad_sg) { /* EMPTY padding behind the last real tree: no walk, the +0 leaves it would end in */ \
          _Pragma("unroll") for (int u = 0; u < U; ++u) lf[0][u] = 0.f;                                \
        } else if constexpr (PIN) {                                                                    \
          if (!slow_l) walk_trees_q16_s2_pin<D, U, TREE_BYTES, FEAT_OFF, false, GL, WIDE>(top_cur, (uint32_t)((BUF) * CHUNK_BYTES + sg * U * TREE_BYTES), lane2, lf[0], gl); \
          else walk_trees_q16_s2_pin<D, U, TREE_BYTES, FEAT_OFF, true, GL, WIDE>(top_cur, (uint32_t)((BUF) * CHUNK_BYTES + sg * U * TREE_BYTES), lane2, lf[0], gl);        \
        } else {                                                                                       \
          if (!slow_l) walk_trees_q16_s2<D, U, TREE_BYTES, FEAT_OFF, false, GL, WIDE>(top_cur, (uint32_t)((BUF) * CHUNK_BYTES + sg * U * TREE_BYTES), lane2, lf[0], gl); \
          else walk_trees_q16_s2<D, U, TREE_BYTES, FEAT_OFF, true, GL, WIDE>(top_cur, (uint32_t)((BUF) * CHUNK_BYTES + sg * U * TREE_BYTES), lane2, lf[0], gl);        \
        }                                                                                              \
      } else if (pad_sg) {                                                                             \
        _Pragma("unroll") for (int u = 0; u < U; ++u) lf[0][u] = 0.f;                                  \
      } else {                                                                                         \
        if (!slow_l) walk_trees_q16<D, U, TREE_BYTES, FEAT_OFF, false, GL, WIDE>((uint32_t)((BUF) * CHUNK_BYTES + sg * U * TREE_BYTES), lane2, lf[0], gl); \
        else walk_trees_q16<D, U, TREE_BYTES, FEAT_OFF, true, GL, WIDE>((uint32_t)((BUF) * CHUNK_BYTES + sg * U * TREE_BYTES), lane2, lf[0], gl);        \
      }                                                                                                \
      if (sum_l != 1) fold_leaves<U, 1, 0>(lf, ((PH) + sg) & 1, C, ra, dacc, exact_l);                 \
      else fold_leaves<U, 1, 1>(lf, ((PH) + sg) & 1, C, ra, dacc);                                     \
      if constexpr (CM || SPLIT) {                                                                     \
        if (U == 8 || (((PH) + sg) & 1) == 1) { /* a PU group is complete */                           \
          if (++cm_groups == cm_bound) { /* ... and it was its cluster's last */                       \
            if constexpr (SPLIT) { /* ... its slice's last / every group: the accumulator goes out as it is */ \
              a.out[(uint64_t)sp_pos * x.n_pad + sp_row] = ra.a[0][0];                                 \
              ra.a[0][0] = 0.f;                                                                        \
              ++sp_pos;                                                                                \
              cm_bound += x.split_len ? 1u : 0x40000000u;                                              \
            } else {                                                                                   \
              cm_total = exact_l ? radd_exact(ra.a[0][0], cm_total) : ra.a[0][0] + cm_total;          \
              ra.a[0][0] = 0.f;                                                                        \
              ++cm_cluster;                                                                            \
              cm_bound += cm_cluster < Cc ? (cm_real + Cc - 1u - cm_cluster) >> cm_lg : 0u;            \
            }                                                                                          \
          }                                                                                            \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
  } while (0)

  // (tried twice, both slower, both removed -- numbers in profiles/archive/r01_tile_overhead.md: levels 0-1 from SGPRs via
  // hidden s_load_dwordx4 of the next chunk's top records, and levels 0-1 tested against ranks kept in 16 VGPRs with
  // a wave-uniform register index; 15 DS ops per tree instead of 17 either way, but the extra wait points cost more
  // than the LDS cycles they save.  Round 3's _s2 form above differs in where the loads are issued and waited for.)
  constexpr int PH1 = (CT == 4) ? 1 : 0;
  // HOT (the _s2 kernels): the common case -- no missing value in the tile, sum_mode 0 -- is dispatched ONCE into a copy of the
  // chunk loop without the per-sub-group branches on `slow` / the sum mode; a chunk is then one basic block and the leaf gathers
  // and adds of one sub-group schedule into the next sub-group's walk: 21.90 vs 23.04 ms at 1000 trees x depth 8 x 20 M tuples
  // (profiles/archive/r03_sweep_q16_hot_dispatch_d8.json; depth 6: equal).  Each path issues its own first record set: a set requested
  // in front of the dispatch would be copied into each path's registers while still in flight (tools/check_s2_isa.py).
  auto chunks = [&](auto hot_tag, auto exact_tag) {
    constexpr bool HOT = decltype(hot_tag)::value, HOT_EXACT = decltype(exact_tag)::value;  // HOT_EXACT: the hot copy for sum_mode 2
    // (deferred folds -- a sub-group's leaves folded one sub-group later, so that the gathers of a chunk's last sub-group fly
    // across the chunk barrier, which then only waits for the DMA: 22.06 vs 21.84 ms, profiles/archive/r03_sweep_q16_deferred_folds.json)
    const bool slow_l = HOT ? false : slow, exact_l = HOT ? HOT_EXACT : exact;
    const int sum_l = HOT ? 0 : SUM1;
    if constexpr (S2) top_issue<TREE_BYTES>(top_a, img);
    for (uint32_t k = 0; k < n_chunks; k += 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const bool more1 = k + 1 < n_chunks;
      if (more1) dma_chunk<THREADS, CHUNK_BYTES>(img + (size_t)(k + 1) * GSKIP, k + 1, CHUNK_BYTES, tid);
      DDT_QCOMPUTE(0, 0, k);
      if (!more1) break;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (k + 2 < n_chunks) dma_chunk<THREADS, CHUNK_BYTES>(img + (size_t)(k + 2) * GSKIP, k + 2, 0, tid);
      DDT_QCOMPUTE(1, PH1, k + 1);
    }
    if constexpr (S2) top_wait(top_a);  // the last request (never used; it went to top_a) must not outlive the wave
  };
  if constexpr (HOTDISP) {
    if (!slow && SUM1 == 0) chunks(std::true_type{}, std::false_type{});
    else if (!slow && SUM1 == 2) chunks(std::true_type{}, std::true_type{});  // the reference adder: its suspect tests are the only branches
    else chunks(std::false_type{}, std::false_type{});
  } else {
    chunks(std::false_type{}, std::false_type{});
  }
#undef DDT_QCOMPUTE
  ra.align(C);
  const uint64_t row = tile0 + (uint64_t)tid;
  if constexpr (CM) {
    if (x.state_out) {  // not the ensemble's last part: the sum's state instead of the score
      x.state_out[row] = ra.a[0][0];
      x.state_out[x.n_pad + row] = cm_total;
      return;
    }
  }
  if constexpr (SPLIT) return;  // (every partial sum went out at its group: [slices or groups][n_pad], whole tiles)
  if (row < a.n) a.out[row] = (SUM1 == 1) ? (float)dacc[0] : CM ? cm_total : ra.total(0, C, exact);
}


template <int D, int CT, int U, int OPT = 0>
static hipError_t launch_q16(const ScoreArgs& a, const Variant& v, hipStream_t s) {
  const Q16Aux& x = *reinterpret_cast<const Q16Aux*>(a.aux);
  const uint64_t tiles = (a.n + kQTile - 1) / kQTile;
  if (tiles == 0) return hipSuccess;
  if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
  auto kern = score_q16_kernel<D, CT, U, OPT>;
  const uint32_t lds = v.lds_bytes_q16(a.tuple_words);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if (!x.skip_prepass) {
    e = launch_q16_prepass(a, x, s);
    if (e != hipSuccess) return e;
  }
  if (a.ev_mid) (void)hipEventRecord(a.ev_mid, s);
  // the cut launch of a small batch (launch_score decides; the same predicate as Variant::has_split): the kernels the automatic choice takes
  if constexpr (D == 8 ? (OPT & 4) != 0 : D >= 5 ? (OPT & 2) != 0 : true) {
    if (x.split > 0u) {  // (slices: clusters, or runs of split_len chunks)
      auto ksplit = score_q16_kernel<D, CT, U, OPT | 128>;
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(ksplit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(ksplit, dim3((uint32_t)tiles, x.split), dim3(kQTile), lds, s, a, x);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL(kern, dim3((uint32_t)tiles), dim3(kQTile), lds, s, a, x);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// "_p": the persistent form of the rank-quantised kernel (always _gl + _s2 + _cm).  What the plain form pays per tile is the
// block turn-over: a new block DMAs its 64 KiB rank tile and its first chunk and only then starts to walk, while the CU's
// other block runs alone at 16 waves -- 9 % of a 125-tree shard's time (16 chunks per tile), 1 % at 1000 trees.  Here a block
// (two per CU, grid = 2 x CUs) stays, and while it walks tile t the rank tile of its NEXT tile sits in 16 VGPRs per lane
// (four coalesced 16-byte loads, issued right after the tile switch): the switch is barrier -> four ds_write_b128 -> barrier.
// With an even chunk count the chunk ring continues across tiles (chunk 0 of the next tile is requested during the last
// chunk, into the buffer that is free then).  Tiles are handed out two ahead: the first two statically (block b: b, b + grid),
// the rest through one atomic counter -- thread 0 takes the ticket during the last chunk and leaves it in the padding word
// (record 0 of tree 0, never read by the _s2 walk) of chunk buffer 1, where every wave finds it behind the tile-end barrier.
// Several ensembles in one image ("segments", Q16Aux): the classes of a one-vs-all model are walked in ONE pass over the tile --
// at a segment's last chunk the lane's sum is stored to out[k][row] and compared with the best so far; the label is written
// once per tuple (BASELINE config 5: K launches + an argmax pass before).  Per class the order of the adds is the reference's
// (FPAddersReduceTree.sv:94-141, FPAggregator.v:79-131, Core.sv:486-541): each segment is a cluster-major image of its own.
// ---------------------------------------------------------------------------------------------------
template <int D, int CT, int U, bool PIN, bool MULTI, bool PROF = false>  // PROF: the diagnostic build of DDT_Q16P_PROFILE (launch_q16p), never the product's
__global__ __launch_bounds__(kQTile, 8) void score_q16p_kernel(const ScoreArgs a, const Q16Aux x) {
  constexpr int THREADS = kQTile;
  constexpr bool GL = true;  // leaves gathered from the global image; levels 0-1 always from SGPRs (_s2)
  static_assert(U == 4 && CT % U == 0 && (CT / U) % 2 == 0 && CT % 8 == 0, "geometry (_s2: an even number of sub-groups per chunk)");
  constexpr int TREE_BYTES = 4 << D;                    // records only: the leaves are gathered from the global image
  constexpr int CHUNK_BYTES = TREE_BYTES * CT;
  constexpr int GCHUNK_UNITS = (8 << D) * CT / 16;
  constexpr int GSKIP = GCHUNK_UNITS - CHUNK_BYTES / 16;
  constexpr int FEAT_OFF = 2 * CHUNK_BYTES;
  constexpr int ROW = kQTile * 2;
  constexpr uint32_t PAD_WORD = (uint32_t)CHUNK_BYTES;  // LDS address of record 0 of tree 0 in chunk buffer 1 (padding: never walked)
  static_assert(FEAT_OFF % ROW == 0, "row|lane OR trick");
  const int tid = threadIdx.x;
  const uint32_t W = a.tuple_words, n_chunks = a.n_chunks;
  const uint32_t seg_chunks = x.seg_chunks ? x.seg_chunks : n_chunks;  // MULTI: chunks per ensemble
  const uint32_t tiles = (uint32_t)(x.n_pad / kQTile), grid = gridDim.x;
  const uint32_t units = W * (ROW / 16);  // 16-byte units of one rank tile
  const bool ring = (n_chunks & 1u) == 0u;
  const uint32_t lane2 = (((uint32_t)tid & 511u) << 2) | (((uint32_t)tid >> 9) << 1);  // see rank_kernel
  const uint32_t Cc = a.clusters, C = 1u;
  const uint32_t cm_lg = (uint32_t)__builtin_ctz(Cc | 0x100u), cm_real = x.real_groups;
  const int SUM1 = (int)a.sum_mode;  // 0 or 2 (the fp64 sum is defined on the stream order: never a _cm image)
  const bool exact = SUM1 == 2;

  // rank tile of tile t -> 4 x 16 bytes per lane; always four loads per wave (units past the tile re-read its last one)
  auto prefetch = [&](u32x4 (&pre)[4], uint32_t t) {
    const u32x4* src = reinterpret_cast<const u32x4*>(x.q + (uint64_t)t * W * kQTile);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t u = (uint32_t)tid + (uint32_t)i * THREADS;
      pre[i] = __builtin_nontemporal_load(src + (u < units ? u : units - 1u));  // read once
    }
  };

  uint32_t cur = blockIdx.x, nxt = cur + grid;  // cur < tiles: the launcher never starts more blocks than tiles
  u32x4 pre[4];
  prefetch(pre, cur);
  bool slow = __builtin_amdgcn_readfirstlane((int)x.tile_flags[cur]) != 0;
  bool pre0 = false;  // chunk 0 of `cur` was requested during the previous tile's last chunk
  TopRecs<4> top_a, top_b;
  // DDT_Q16P_PROFILE: where a tile's time goes, by wave 0's 100 MHz clock: [0] tile switch (loop top -> rank tile in LDS, next tile's
  // ranks requested), [1] -> first chunk barrier passed, [2] -> last chunk walked, [3] -> tile-end barrier passed, [4] tiles
  unsigned long long pt[6] = {0, 0, 0, 0, 0, 0};  // [5]: the first two chunks of the tile (the rest of the walk stays in [2])
  unsigned long long t_mark = (PROF && x.dbg) ? __builtin_amdgcn_s_memrealtime() : 0ull;
  // ... and per chunk index k < 128: [0] barrier passed -> chunk k walked, [1] chunk k-1 walked -> barrier of chunk k passed (global atomics)
  unsigned long long t_chunk = (PROF && x.dbg) ? __builtin_amdgcn_s_memrealtime() : 0ull;
  auto lapk = [&](uint32_t k, int which) {
    if (PROF && x.dbg) {  // (the run-time test on top of the compile-time one: without it hipcc fails with "illegal VGPR to SGPR copy")
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (tid == 0 && k < 128u) atomicAdd(x.dbg + (size_t)gridDim.x * 8u + (size_t)which * 128u + k, now - t_chunk);
      t_chunk = now;
    }
  };
  auto lap = [&](int i) {
    if (PROF && x.dbg) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      pt[i] += now - t_mark;
      t_mark = now;
    }
  };

  for (;;) {
    const uint4* img = slow ? x.img_slow : a.img;
    const bool has_next = nxt < tiles;
    // the next tile's flag, needed at this tile's last chunk (which image its chunk 0 comes from)
    // (this load is waited for on the spot; without it the shard measured the same 13.04 ms: profiles/r04_raw/r04_s9)
    const bool slow_n = has_next && __builtin_amdgcn_readfirstlane((int)x.tile_flags[has_next ? nxt : cur]) != 0;
    const uint4* img_n = slow_n ? x.img_slow : a.img;
    if (!pre0) dma_chunk<THREADS, CHUNK_BYTES>(img, 0, 0, tid);
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // every wave is behind the tile-end barrier: the old tile is dead
      const uint32_t u = (uint32_t)tid + (uint32_t)i * THREADS;
      if (u < units) *reinterpret_cast<DDT_LDS(u32x4)*>((uint32_t)FEAT_OFF + u * 16u) = pre[i];
    }
    prefetch(pre, has_next ? nxt : cur);  // flies during this tile's walks
    lap(0);

    const __amdgpu_buffer_rsrc_t leaf_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(img), 0, (int)(n_chunks * (uint32_t)(GCHUNK_UNITS * 16)), 0x00020000);
    RefAcc<1> ra;
    ra.init();
    double dacc[1] = {0.0};
    uint32_t cm_groups = 0, cm_cluster = 0, cm_bound = (cm_real + Cc - 1u) >> cm_lg;
    float cm_total = 0.f;
    uint32_t seg = 0, seg_left = seg_chunks;
    float best = 0.f;
    int32_t arg = 0;
    uint32_t ticket = 0;
    const uint64_t row = (uint64_t)cur * kQTile + (uint64_t)tid;

#define DDT_QPCOMPUTE(BUF, KIDX)                                                                       \
  do {                                                                                                 \
    _Pragma("unroll") for (int sg = 0; sg < CT / U; ++sg) {                                            \
      float lf[1][U];                                                                                  \
      const LeafSrc gl = {leaf_rsrc, (uint32_t)(KIDX) * (uint32_t)(GCHUNK_UNITS * 16) + (uint32_t)((CT + sg * U - 1) * (4 << D))}; \
      const uint32_t kn = (sg + 1 < CT / U) ? (uint32_t)(KIDX) : ((uint32_t)(KIDX) + 1u < n_chunks ? (uint32_t)(KIDX) + 1u : 0u); \
      const int sn = (sg + 1 < CT / U) ? sg + 1 : 0;                                                   \
      TopRecs<4>& top_cur = (sg & 1) ? top_b : top_a;                                                  \
      TopRecs<4>& top_nxt = (sg & 1) ? top_a : top_b;                                                  \
      top_wait(top_cur);                                                                               \
      top_issue<TREE_BYTES>(top_nxt, img + (size_t)kn * GCHUNK_UNITS + (size_t)(sn * U) * (TREE_BYTES / 16)); \
      if (MULTI && sg == CT / U - 1 && seg_left == x.seg_tail_left) {                                 \
        /* the second half of the ensemble's partly filled PU group is EMPTY padding (100 trees per class): no walk, their +0 leaves */ \
        _Pragma("unroll") for (int u = 0; u < U; ++u) lf[0][u] = 0.f;                                  \
      } else if constexpr (PIN) {                                                                      \
        if (!slow_l) walk_trees_q16_s2_pin<D, U, TREE_BYTES, FEAT_OFF, false, GL>(top_cur, (uint32_t)((BUF) * CHUNK_BYTES + sg * U * TREE_BYTES), lane2, lf[0], gl); \
        else walk_trees_q16_s2_pin<D, U, TREE_BYTES, FEAT_OFF, true, GL>(top_cur, (uint32_t)((BUF) * CHUNK_BYTES + sg * U * TREE_BYTES), lane2, lf[0], gl);        \
      } else {                                                                                         \
        if (!slow_l) walk_trees_q16_s2<D, U, TREE_BYTES, FEAT_OFF, false, GL>(top_cur, (uint32_t)((BUF) * CHUNK_BYTES + sg * U * TREE_BYTES), lane2, lf[0], gl); \
        else walk_trees_q16_s2<D, U, TREE_BYTES, FEAT_OFF, true, GL>(top_cur, (uint32_t)((BUF) * CHUNK_BYTES + sg * U * TREE_BYTES), lane2, lf[0], gl);        \
      }                                                                                                \
      fold_leaves<U, 1, 0>(lf, sg & 1, C, ra, dacc, exact_l);                                          \
      if ((sg & 1) == 1) { /* a PU group is complete */                                                \
        if (++cm_groups == cm_bound) { /* ... and it was its cluster's last */                         \
          cm_total = exact_l ? radd_exact(ra.a[0][0], cm_total) : ra.a[0][0] + cm_total;              \
          ra.a[0][0] = 0.f;                                                                            \
          ++cm_cluster;                                                                                \
          cm_bound += cm_cluster < Cc ? (cm_real + Cc - 1u - cm_cluster) >> cm_lg : 0u;                \
        }                                                                                              \
      }                                                                                                \
    }                                                                                                  \
    if constexpr (MULTI) {                                                                             \
      if (--seg_left == 0u) { /* wave-uniform: the ensemble ends with this chunk */                    \
        if (a.out && row < a.n) a.out[(uint64_t)seg * a.n + row] = cm_total;                           \
        if (seg == 0u || cm_total > best || (best != best && cm_total == cm_total)) {                  \
          best = cm_total;                                                                             \
          arg = (int32_t)seg;                                                                          \
        }                                                                                              \
        ++seg;                                                                                         \
        seg_left = seg_chunks;                                                                         \
        ra.a[0][0] = 0.f;                                                                              \
        cm_groups = 0u;                                                                                \
        cm_cluster = 0u;                                                                               \
        cm_bound = (cm_real + Cc - 1u) >> cm_lg;                                                       \
        cm_total = 0.f;                                                                                \
      }                                                                                                \
    }                                                                                                  \
  } while (0)

    // the tile's last chunk: thread 0 takes the block's ticket for the tile after next (begin), and leaves it in the padding
    // word of chunk buffer 1 (end) -- that buffer is the live one (ring) or dead since the previous chunk barrier (odd chunk
    // count), and the next DMA into it is only issued behind the barrier after the tile-end barrier
    auto last_begin = [&]() {
      // (static assignment instead -- tile + 2 * grid, no atomic -- measured 14.72 vs 13.04 ms on a 125-tree shard: the blocks do not run
      // at one speed, and a ticket counter is what the hardware dispatcher gives the plain launch; profiles/r04_raw/r04_s9)
      if (tid == 0) ticket = has_next ? 2u * grid + atomicAdd(x.tile_counter, 1u) : 0xFFFFFFFFu;
    };
    auto last_end = [&]() {
      if (tid == 0) lds_st_u32(PAD_WORD, ticket);
    };

    auto chunks = [&](auto hot_tag, auto exact_tag) {
      constexpr bool HOT = decltype(hot_tag)::value, HOT_EXACT = decltype(exact_tag)::value;
      const bool slow_l = HOT ? false : slow, exact_l = HOT ? HOT_EXACT : exact;
      top_issue<TREE_BYTES>(top_a, img);
      for (uint32_t k = 0; k < n_chunks; k += 2) {
        // (k == 0 behind the ring: chunk 0 arrived before the tile-end barrier; a vmcnt(0) here would only expose the HBM latency of the
        // next tile's ranks, requested a moment ago -- 4 us per tile)
        if (k != 0u || !pre0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // chunk k is in buffer 0 (k = 0: and the rank tile is in place); everyone is done with buffer 1
        if (k == 0u) lap(1);
        lapk(k, 1);
        const bool more1 = k + 1 < n_chunks;
        if (more1) dma_chunk<THREADS, CHUNK_BYTES>(img + (size_t)(k + 1) * GSKIP, k + 1, CHUNK_BYTES, tid);
        else last_begin();
        DDT_QPCOMPUTE(0, k);
        lapk(k, 0);
        if (!more1) {
          last_end();
          break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        lapk(k + 1u, 1);
        const bool more2 = k + 2 < n_chunks;
        if (more2) dma_chunk<THREADS, CHUNK_BYTES>(img + (size_t)(k + 2) * GSKIP, k + 2, 0, tid);
        else {
          if (has_next) dma_chunk<THREADS, CHUNK_BYTES>(img_n, 0, 0, tid);  // ring: chunk 0 of the next tile
          last_begin();
        }
        DDT_QPCOMPUTE(1, k + 1);
        lapk(k + 1u, 0);
        if (k == 0u) lap(5);
        if (!more2) last_end();
      }
      top_wait(top_a);  // the last request (never used; it went to top_a) must not outlive this pass
    };
    if (!slow && SUM1 == 0) chunks(std::true_type{}, std::false_type{});
    else if (!slow && SUM1 == 2) chunks(std::true_type{}, std::true_type{});
    else chunks(std::false_type{}, std::false_type{});
#undef DDT_QPCOMPUTE
    if constexpr (MULTI) {
      if (x.labels && row < a.n) x.labels[row] = arg;
    } else {
      if (row < a.n) a.out[row] = cm_total;
    }
    lap(2);
    if (PROF && x.dbg) ++pt[4];
    if (!has_next) break;
    // tile end: every walk of this tile is done (rank tile, chunk buffers), the ticket is in place -- and whatever this wave has
    // requested has arrived (the ring's chunk 0 of the next tile, requested a chunk ago): the next tile's first barrier then needs no
    // vmcnt wait and the ranks requested right behind this barrier stay in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lap(3);
    const uint32_t nn = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_u32(PAD_WORD));
    cur = nxt;
    nxt = nn;
    slow = slow_n;
    pre0 = ring;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing requested may outlive the wave
  if (PROF && x.dbg && tid == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) x.dbg[(size_t)blockIdx.x * 8u + (size_t)i] = pt[i];
  }
}

template <int D, int CT, int U, bool PIN>
static hipError_t launch_q16p(const ScoreArgs& a, const Variant& v, hipStream_t s) {
  const bool multi = reinterpret_cast<const Q16Aux*>(a.aux)->n_segs > 1u;
  Q16Aux x = *reinterpret_cast<const Q16Aux*>(a.aux);
  const uint64_t tiles = (a.n + kQTile - 1) / kQTile;
  if (tiles == 0) return hipSuccess;
  if (tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
  if (x.n_segs == 0u || (x.seg_chunks ? x.seg_chunks * x.n_segs : a.n_chunks) != a.n_chunks || a.sum_mode == 1u) return hipErrorInvalidValue;
  x.tile_counter = x.tile_flags + ((tiles + 1u) & ~(uint64_t)1u) + 2u * kQ16GroupedCounters;  // behind the pre-pass counters (launch_q16_prepass)
  auto kern = multi ? score_q16p_kernel<D, CT, U, PIN, true> : score_q16p_kernel<D, CT, U, PIN, false>;  // several ensembles in the image: sums + argmax
  const uint32_t lds = v.lds_bytes_q16(a.tuple_words);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if (!x.skip_prepass) {
    e = launch_q16_prepass(a, x, s);  // its memset also zeroes the tile counter
    if (e != hipSuccess) return e;
  } else {
    e = launch_zero_words(x.tile_counter, kQ16TileCounterWords, s);
    if (e != hipSuccess) return e;
  }
  if (a.ev_mid) (void)hipEventRecord(a.ev_mid, s);
  uint64_t grid = 2ull * a.num_cus;  // two resident blocks per CU (LDS: 2 x 80 KiB at 32 words per tuple; 8 waves per SIMD)
  if (grid > tiles) grid = tiles;
  if (getenv("DDT_Q16P_PROFILE")) {  // diagnostic: synchronous, prints the phase sums of this launch (see the kernel)
    unsigned long long* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), grid * 64 + 2048) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(d, 0, grid * 64 + 2048, s);
    x.dbg = d;
    auto pkern = score_q16p_kernel<D, CT, U, PIN, false, true>;
    if (multi) return hipErrorInvalidValue;  // (one ensemble per launch only)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(pkern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pkern, dim3((uint32_t)grid), dim3(kQTile), lds, s, a, x);
    std::vector<unsigned long long> h(grid * 8 + 256);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), d, grid * 64 + 2048, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    double sum[6] = {0, 0, 0, 0, 0, 0}, mx_total = 0;
    for (uint64_t b = 0; b < grid; ++b) {
      double tot = 0;
      for (int i = 0; i < 6; ++i) sum[i] += (double)h[b * 8 + i];
      tot = (double)(h[b * 8] + h[b * 8 + 1] + h[b * 8 + 2] + h[b * 8 + 3] + h[b * 8 + 5]);
      mx_total = tot > mx_total ? tot : mx_total;
    }
    const double tl = sum[4] > 0 ? sum[4] : 1;
    fprintf(stderr, "{\"q16p_profile\": {\"blocks\": %llu, \"tiles\": %.0f, \"chunks_per_tile\": %u, \"us_per_tile\": {\"switch\": %.3f, \"to_first_barrier\": %.3f, "
                    "\"first_two_chunks\": %.3f, \"other_chunks\": %.3f, \"tile_end_barrier\": %.3f}, \"slowest_block_ms\": %.3f, \"mean_block_ms\": %.3f}}\n",
            (unsigned long long)grid, sum[4], a.n_chunks, sum[0] / tl * 0.01, sum[1] / tl * 0.01, sum[5] / tl * 0.01, sum[2] / tl * 0.01, sum[3] / tl * 0.01, mx_total * 1e-5,
            (sum[0] + sum[1] + sum[2] + sum[3] + sum[5]) / (double)grid * 1e-5);
    fprintf(stderr, "{\"q16p_chunk_us\": {\"walk\": [");
    for (uint32_t k = 0; k < a.n_chunks && k < 128u; ++k) fprintf(stderr, "%s%.3f", k ? ", " : "", (double)h[grid * 8 + k] / tl * 0.01);
    fprintf(stderr, "], \"wait_before\": [");
    for (uint32_t k = 0; k < a.n_chunks && k < 128u; ++k) fprintf(stderr, "%s%.3f", k ? ", " : "", (double)h[grid * 8 + 128 + k] / tl * 0.01);
    fprintf(stderr, "]}}\n");
    return hipGetLastError();
  }
  hipLaunchKernelGGL(kern, dim3((uint32_t)grid), dim3(kQTile), lds, s, a, x);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// generic kernel: any D (1..16), any F (1..2048).  Lane = tuple, 256 tuples per block, one tree at a
// time.  Features in LDS when the tile fits, else gathered from global memory; tree in LDS when it
// fits (12*2^D bytes), else nodes are read from global memory (L2).  Correctness path for shapes the
// specialised kernels do not cover; same image format with w2 = feature index | miss_right<<31.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kGenericLdsBudget = 150u * 1024u;

uint32_t generic_lds_bytes(uint32_t levels, uint32_t tuple_words, bool* feat_in_lds, bool* tree_in_lds, uint32_t* top_levels) {
  const uint64_t feat = (uint64_t)tuple_words * kGenericThreads * 4u;
  const uint64_t tree = 12ull << levels;
  bool f = feat <= 96u * 1024u;
  // whole tree in LDS only while it is small: from depth 9 on, re-staging 12*2^D bytes per tree and block costs more
  // than gathering the lower levels (measured 64 x d12: 3.5 ms vs the staged-top form below)
  bool t = levels <= 8u && tree + (f ? feat : 0) <= kGenericLdsBudget;
  // deep trees (config 4): the top levels of the 8 trees of a PU group are staged in LDS, only the lower levels are
  // gathered from L2/HBM.  As many levels as keep two blocks per CU (80 KiB each), at least 6 if the block fits at all.
  uint32_t top = 0;
  if (!t) {
    const uint64_t base = f ? feat : 0;
    for (uint32_t d = 1; d < levels && d <= 10u; ++d) {
      const uint64_t need = base + 8ull * (8ull << d);
      if (need <= 80u * 1024u || (d <= 6u && need <= kGenericLdsBudget)) top = d;
    }
  }
  if (feat_in_lds) *feat_in_lds = f;
  if (tree_in_lds) *tree_in_lds = t;
  if (top_levels) *top_levels = top;
  return (uint32_t)((f ? feat : 0) + (t ? tree : 0) + (top ? 8ull * (8ull << top) : 0));
}

template <bool FEAT_LDS, bool TREE_LDS>
__global__ __launch_bounds__(kGenericThreads) void score_generic_kernel(const ScoreArgs a) {
  constexpr uint32_t TILE = kGenericThreads;
  const uint32_t tid = threadIdx.x;
  const uint32_t W = a.tuple_words, D = a.levels;
  const uint64_t row = (uint64_t)blockIdx.x * TILE + tid;
  const bool valid = row < a.n;
  const uint32_t* xrow = a.tuples + (valid ? row : 0) * W;
  const uint32_t tree_bytes = 12u << D;
  const uint32_t tree_lds = FEAT_LDS ? W * TILE * 4u : 0u;  // LDS: [features][one tree]

  if (FEAT_LDS) {
    for (uint32_t j = 0; j < W; ++j) {
      uint32_t v = valid ? xrow[j] : 0u;
      if (a.ieee) v = (v == a.miss_raw) ? kMissSentinelIeee : ieee_key(v);
      lds_st_u32((j * TILE + tid) * 4u, v);
    }
  }
  RefAcc<1> ra;
  ra.init();
  double dacc = 0.0;
  float grp[8];
  for (uint32_t t8 = 0; t8 < a.n_trees; t8 += 8u) {
    if (!TREE_LDS) {
      // deep trees: the top levels of the group's 8 trees are staged in LDS, the lower ones gathered from L2 / HBM;
      // the 8 trees advance level by level together, so a lane has 8 independent node reads in flight
      const unsigned char* base = reinterpret_cast<const unsigned char*>(a.img) + (size_t)t8 * tree_bytes;
      const uint32_t DT = a.top_levels, top_bytes = 8u << DT;  // per tree: heap records 0 .. 2^DT - 1
      if (DT) {
        __syncthreads();  // previous group's top levels fully consumed
        for (uint32_t off = tid * 16u; off < 8u * top_bytes; off += TILE * 16u) {
          const uint32_t tu = off / top_bytes, o = off - tu * top_bytes;
          lds_st_u4(tree_lds + off, *reinterpret_cast<const uint4*>(base + (size_t)tu * tree_bytes + o));
        }
        __syncthreads();
      }
      uint32_t m[8];
#pragma unroll
      for (uint32_t tu = 0; tu < 8u; ++tu) m[tu] = 1u;
      for (uint32_t lvl = 0; lvl < D; ++lvl) {
        uint2 nd[8];
        if (lvl < DT) {  // wave-uniform
#pragma unroll
          for (uint32_t tu = 0; tu < 8u; ++tu) nd[tu] = lds_u2(tree_lds + tu * top_bytes + m[tu] * 8u);
        } else {
#pragma unroll
          for (uint32_t tu = 0; tu < 8u; ++tu)
            nd[tu] = *reinterpret_cast<const uint2*>(base + (size_t)tu * tree_bytes + (size_t)m[tu] * 8u);
        }
#pragma unroll
        for (uint32_t tu = 0; tu < 8u; ++tu) {
          const uint32_t j = nd[tu].y & 0x7FFFFFFFu;
          uint32_t f;
          if (FEAT_LDS) f = lds_u32((j * TILE + tid) * 4u);
          else {
            f = xrow[j];
            if (a.ieee) f = (f == a.miss_raw) ? kMissSentinelIeee : ieee_key(f);
          }
          m[tu] = 2u * m[tu] + (go_right<true>(f, nd[tu].x, nd[tu].y, a.miss_key) ? 1u : 0u);
        }
      }
#pragma unroll
      for (uint32_t tu = 0; tu < 8u; ++tu) {
        const uint32_t lo = (8u << D) + (m[tu] - (1u << D)) * 4u;
        grp[tu] = *reinterpret_cast<const float*>(base + (size_t)tu * tree_bytes + lo);
      }
      if (a.sum_mode == 1) {
#pragma unroll
        for (uint32_t tu = 0; tu < 8u; ++tu) dacc += (double)grp[tu];
      }
    } else {
#pragma unroll
      for (uint32_t tu = 0; tu < 8u; ++tu) {
        const uint32_t t = t8 + tu;
        const unsigned char* timg = reinterpret_cast<const unsigned char*>(a.img) + (size_t)t * tree_bytes;
        if (TREE_LDS) {
          __syncthreads();  // previous tree fully consumed
          for (uint32_t off = tid * 16u; off < tree_bytes; off += TILE * 16u)
            lds_st_u4(tree_lds + off, *reinterpret_cast<const uint4*>(timg + off));
          __syncthreads();
        }
        uint32_t m = 1;
        for (uint32_t lvl = 0; lvl < D; ++lvl) {
          const uint2 nd = TREE_LDS ? lds_u2(tree_lds + m * 8u) : *reinterpret_cast<const uint2*>(timg + (size_t)m * 8u);
          const uint32_t j = nd.y & 0x7FFFFFFFu;
          uint32_t f;
          if (FEAT_LDS) f = lds_u32((j * TILE + tid) * 4u);
          else {
            f = xrow[j];
            if (a.ieee) f = (f == a.miss_raw) ? kMissSentinelIeee : ieee_key(f);
          }
          m = 2u * m + (go_right<true>(f, nd.x, nd.y, a.miss_key) ? 1u : 0u);
        }
        const uint32_t lo = (8u << D) + (m - (1u << D)) * 4u;
        const float leaf = TREE_LDS ? lds_f32(tree_lds + lo) : *reinterpret_cast<const float*>(timg + lo);
        if (a.sum_mode == 1) dacc += (double)leaf;
        grp[tu] = leaf;
      }
    }
    if (a.sum_mode != 1) {
      const float lf[1][8] = {{grp[0], grp[1], grp[2], grp[3], grp[4], grp[5], grp[6], grp[7]}};
      double unused[1] = {0.0};
      fold_leaves<8, 1, 0>(lf, 0, a.clusters, ra, unused, a.sum_mode == 2);
    }
  }
  ra.align(a.clusters);
  if (valid) a.out[row] = (a.sum_mode == 1) ? (float)dacc : ra.total(0, a.clusters, a.sum_mode == 2);
}

hipError_t launch_generic(const ScoreArgs& a_in, const Variant&, hipStream_t s) {
  bool fl, tl;
  ScoreArgs a = a_in;
  const uint32_t lds = generic_lds_bytes(a.levels, a.tuple_words, &fl, &tl, &a.top_levels);
  const uint64_t blocks = (a.n + kGenericThreads - 1) / kGenericThreads;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
#define DDT_GEN(FL, TL)                                                                                              \
  do {                                                                                                               \
    auto kern = score_generic_kernel<FL, TL>;                                                                        \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                                                   \
    hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(kGenericThreads), lds, s, a);                              \
  } while (0)
  if (fl && tl) DDT_GEN(true, true);
  else if (fl) DDT_GEN(true, false);
  else if (tl) DDT_GEN(false, true);
  else DDT_GEN(false, false);
#undef DDT_GEN
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// chain sum of partial score vectors: out = (((p0 + p1) + p2) + ...)  (ResultsCombiner.sv:292-311)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chain_sum_kernel(const float* __restrict__ parts, uint32_t n_parts, size_t n, size_t pitch,
                                                        float* __restrict__ out, const bool exact) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float run = parts[i];
    if (!exact) {
      for (uint32_t p = 1; p < n_parts; ++p) run = parts[(size_t)p * pitch + i] + run;  // local + upstream
    } else {
      for (uint32_t p = 1; p < n_parts; ++p) run = radd_exact(parts[(size_t)p * pitch + i], run);  // sum_mode 2: the hop adders are the same FloPoCo adder
    }
    out[i] = run;
  }
}

// The adds of a launch cut into slices (score_q16_kernel SPLIT), in the reference's order: per cluster acc <- p + acc over its partial sums
// (FPAggregator.v:79-131; one per PU group in image order, or the cluster's finished accumulator), then total <- acc + total over the
// clusters (Core.sv:486-541).  parts = [positions][pitch]; cluster c holds (real + C - 1 - c) / C groups (ddt_image.cpp cm_position); position = the
// group's place in the image the scoring kernel walked: cluster by cluster (cm_order) or in stream order.
// A block = 64 tuples x 4 waves.  The clusters' chains are independent of each other: wave q runs the chains of clusters q, q + 4 -- sixteen
// independent, coalesced loads in flight, then their adds in order -- and leaves the accumulators in LDS; wave 0 adds them in cluster order.
// (One thread per tuple walking 125 dependent global loads took 31 us for a single tile; staged through LDS by a loop hipcc did not unroll 16.)
__global__ __launch_bounds__(256) void cm_combine_kernel(const float* __restrict__ parts, size_t pitch, size_t n, uint32_t real_groups, uint32_t clusters,
                                                         uint32_t per_group, uint32_t cm_order, float* __restrict__ out, const bool exact,
                                                         uint32_t class_positions, size_t out_pitch) {
  // blockIdx.y = class of a one-vs-all model whose classes stand back to back in the image (class_positions partial sums each): its sum to out[class][row]
  parts += (size_t)blockIdx.y * class_positions * pitch;
  out += (size_t)blockIdx.y * out_pitch;
  extern __shared__ float cacc_dyn[];  // [8][64]: clusters_per_tuple is 1, 2, 4 or 8 (ddt_model.cpp); dynamic like every LDS byte of this library (tests/test_abi_host.py)
  float (*cacc)[64] = reinterpret_cast<float (*)[64]>(cacc_dyn);
  const uint32_t r = threadIdx.x & 63u, q = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * 64u + r;  // (< pitch: the partial vectors are whole tiles)
  auto count_of = [&](uint32_t c) -> uint32_t {     // partial sums of cluster c
    const uint32_t len = (real_groups + clusters - 1u - c) / clusters;
    return per_group ? len : (len ? 1u : 0u);
  };
  uint32_t pos = 0;
  for (uint32_t c = 0; c < clusters; ++c) {  // (wave-uniform)
    const uint32_t cnt = count_of(c);
    if ((c & 3u) == q) {
      float acc = 0.f;
      // the cluster's j-th partial sum: at pos + j in a cluster-major image; in stream order group g belongs to cluster g mod C (Core.sv:291-316)
      const float* src = parts + (size_t)(cm_order ? pos : c) * pitch + row;
      const size_t step = (size_t)(cm_order ? 1u : clusters) * pitch;
      for (uint32_t j0 = 0; j0 < cnt; j0 += 16u) {
        float v[16];
#pragma unroll
        for (uint32_t u = 0; u < 16u; ++u) v[u] = j0 + u < cnt ? src[(size_t)(j0 + u) * step] : 0.f;
#pragma unroll
        for (uint32_t u = 0; u < 16u; ++u)
          if (j0 + u < cnt) acc = exact ? radd_exact(v[u], acc) : v[u] + acc;
      }
      cacc[c][r] = acc;
    }
    pos += cnt;
  }
  __syncthreads();
  if (q == 0u && row < n) {
    float total = 0.f;
    for (uint32_t c = 0; c < clusters; ++c)
      if (count_of(c)) total = exact ? radd_exact(cacc[c][r], total) : cacc[c][r] + total;
    out[row] = total;
  }
}

hipError_t launch_cm_combine(const float* parts, size_t pitch, size_t n, uint32_t real_groups, uint32_t clusters, bool per_group, bool cm_order, float* out,
                             bool exact, hipStream_t s, uint32_t n_classes, uint32_t class_positions, size_t out_pitch) {
  if (n == 0) return hipSuccess;
  (void)hipGetLastError();
  const size_t blocks = (n + 63) / 64;
  if (blocks > 0x7FFFFFFFull || n_classes == 0u || n_classes > 65535u) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cm_combine_kernel, dim3((uint32_t)blocks, n_classes), dim3(256), 8 * 64 * sizeof(float), s, parts, pitch, n, real_groups, clusters, per_group ? 1u : 0u,
                     cm_order ? 1u : 0u, out, exact, class_positions, out_pitch);
  return hipGetLastError();
}

// pitch = elements between consecutive partial vectors (0: n, the vectors stand back to back)
hipError_t launch_chain_sum(const float* parts, uint32_t n_parts, size_t n, float* out, bool exact, hipStream_t s, size_t pitch) {
  if (n == 0) return hipSuccess;
  (void)hipGetLastError();  // do not inherit a stale error
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048 * 8) blocks = 2048 * 8;
  hipLaunchKernelGGL(chain_sum_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, parts, n_parts, n, pitch ? pitch : n, out, exact);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// argmax over K per-class score vectors laid out [K][n]; lowest class index wins ties; a NaN score never
// wins against a number (BASELINE config 5: one-vs-all, per-class sums then argmax -- an extension, the
// reference has no classes)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ scores, uint32_t K, size_t pitch, size_t n,
                                                     int32_t* __restrict__ labels) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float best = scores[i];
    int32_t arg = 0;
    for (uint32_t k = 1; k < K; ++k) {
      const float s = scores[(size_t)k * pitch + i];
      if (s > best || (best != best && s == s)) {
        best = s;
        arg = (int32_t)k;
      }
    }
    labels[i] = arg;
  }
}

// n columns of a [K][pitch] block (a row group's slice of the class sums of a hybrid job: pitch = rows of the whole batch)
hipError_t launch_argmax_strided(const float* scores, uint32_t K, size_t pitch, size_t n, int32_t* labels, hipStream_t s) {
  if (n == 0) return hipSuccess;
  (void)hipGetLastError();  // do not inherit a stale error
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048 * 8) blocks = 2048 * 8;
  hipLaunchKernelGGL(argmax_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, scores, K, pitch, n, labels);
  return hipGetLastError();
}
hipError_t launch_argmax(const float* scores, uint32_t K, size_t n, int32_t* labels, hipStream_t s) { return launch_argmax_strided(scores, K, n, n, labels, s); }

// ---------------------------------------------------------------------------------------------------
// synthetic tuple generator: x[r][j] = unit(splitmix64(SEED_X + r*F + j))
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void synth_tuples_kernel(uint32_t* __restrict__ out, uint64_t row0, size_t n, uint32_t F,
                                                           uint32_t W, int dist, uint32_t missing_bits) {
  const size_t total = n * (size_t)W;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t r = i / W;
    const uint32_t j = (uint32_t)(i - r * W);
    uint32_t bits = 0u;
    if (j < F) {
      const uint64_t h = splitmix64(kSeedX + (row0 + r) * (uint64_t)F + j);
      float v = (float)(h >> 40) * (1.0f / 16777216.0f);
      if (dist == 1) {
        v = v * 2.0f - 1.0f;
        bits = (((h >> 8) & 0xFFFFull) % 20ull == 0ull) ? missing_bits : __float_as_uint(v);
      } else {
        bits = __float_as_uint(v);
      }
    }
    out[i] = bits;
  }
}

hipError_t launch_synth_tuples(uint32_t* out, uint64_t row0, size_t n, uint32_t F, int dist, uint32_t missing_bits,
                               hipStream_t s) {
  if (n == 0) return hipSuccess;
  (void)hipGetLastError();  // do not inherit a stale error
  const uint32_t W = (F + 3u) / 4u * 4u;
  hipLaunchKernelGGL(synth_tuples_kernel, dim3(256 * 16), dim3(256), 0, s, out, row0, n, F, W, dist, missing_bits);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// variant table
// ---------------------------------------------------------------------------------------------------
#define DDT_V(NAME, D, TH, R, CT, U, ST, OPT) \
  Variant { NAME, kKindTile, D, TH, R, CT, U, ST, OPT, &launch_tile<D, TH, R, CT, U, ST, OPT> }
#define DDT_S(NAME, D, U, MAXLPT) \
  Variant { NAME, kKindStream, D, kStreamThreads, 1, 8, U, 0, MAXLPT, &launch_stream<D, U, MAXLPT> }

#define DDT_Q(NAME, D, CT, U) \
  Variant { NAME, kKindQ16, D, kQTile, 1, CT, U, 1, 0, &launch_q16<D, CT, U> }
#define DDT_QG(NAME, D, CT, U) /* leaves gathered from global memory: only the node records are staged, twice the trees per chunk */ \
  Variant { NAME, kKindQ16, D, kQTile, 1, CT, U, 1, 1, &launch_q16<D, CT, U, 1> }
#define DDT_QGS(NAME, D, CT, U) /* _gl + levels 0-1 from SGPRs (scalar loads) */ \
  Variant { NAME, kKindQ16, D, kQTile, 1, CT, U, 1, 3, &launch_q16<D, CT, U, 3> }
#define DDT_QO(NAME, D, CT, U, OPT) /* any q16 option set: bit 0 _gl, bit 1 _s2 */ \
  Variant { NAME, kKindQ16, D, kQTile, 1, CT, U, 1, OPT, &launch_q16<D, CT, U, OPT> }

static const Variant g_variants[] = {
    Variant{"generic", kKindGeneric, 0, kGenericThreads, 1, 1, 1, 0, 0, &launch_generic},
    // rank-quantised u16 path: 2 blocks x 1024 threads per CU
    // (`q16_d8_c4_u4`, nodes AND leaves of four trees per chunk in LDS, went at the end of round 6: unreachable -- `_gl` fits wherever it fitted)
    // _gl: the leaves stay in global memory (one 4-byte gather per tree on the otherwise idle vector-memory path), only the node
    // records are staged: the most conflict-laden LDS read of a tree (256 leaves behind 32 banks) is gone and a chunk holds 8
    // trees, so half the barriers.  1000 trees x 50 M tuples: 56.4 vs 59.4 ms; 8 trees in flight per lane (u8): 58.6
    DDT_QG("q16_d8_c8_u4_gl", 8, 8, 4),
    DDT_QGS("q16_d8_c8_u4_gl_s2", 8, 8, 4),
    // _cm: cluster-major image order, one accumulator + a running total instead of the ring of C accumulators (sum modes 0 and 2) -- the forms that
    // have it are `_p` and `_x` (`q16_d8_c8_u4_gl_s2_cm` itself, the unpinned walk, went at the end of round 6: `_x` replaced it everywhere)
    // _p: persistent blocks, the next rank tile prefetched into registers, several ensembles (classes) per pass (opt bit 3)
    Variant{"q16_d8_c8_u4_gl_s2_cm_p", kKindQ16, 8, kQTile, 1, 8, 4, 1, 15, &launch_q16p<8, 8, 4, true>},
    // _x: the plain launch with the pinned read order (four chains in flight per lane)
    Variant{"q16_d8_c8_u4_gl_s2_cm_x", kKindQ16, 8, kQTile, 1, 8, 4, 1, 7, &launch_q16<8, 8, 4, 23>},
    // _s2 on the layouts that keep their leaves in LDS (depths 5-7): 100 x d6 x 28 features, 10 M tuples: 1.297 vs 1.333 ms
    DDT_QO("q16_d6_c16_u4_s2", 6, 16, 4, 2),
    DDT_QO("q16_d7_c8_u4_s2", 7, 8, 4, 2),
    DDT_QO("q16_d5_c32_u4_s2", 5, 32, 4, 2),
    // (the pinned read order of the depth-8 "_x" kernel does not pay at depth 6: 0.896 vs 0.876 ms per 10 M tuples x 100 trees,
    // gpurun_out r04_s6 -- with the leaves in LDS and six levels hipcc's own order is the better one; not instantiated)
    DDT_Q("q16_d6_c16_u4", 6, 16, 4),
    DDT_Q("q16_d4_c64_u8", 4, 64, 8),
    // odd depths (XGBoost / scikit-learn defaults 3, 5, 7): same 8 KiB chunks
    DDT_Q("q16_d7_c8_u4", 7, 8, 4),
    DDT_Q("q16_d5_c32_u4", 5, 32, 4),
    DDT_Q("q16_d3_c128_u8", 3, 128, 8),
    // (the one-block depth 9 / 10 forms "q16_d9_c4_u4" / "q16_d10_c4_u4" went in round 6: the deep kernels of ddt_deep.hip replaced them in round 5,
    // 2418 / 2125 against 1823 / 1628 Mtuples/s; so did the persistent fp32 tile kernel "d8_t1024_r1_c4_u4_dma_fp", which no choice ever took)
    // tuples of 33..64 words ("q16w" / "q16dw": the record carries half the row offset, one block of 16 waves per CU: 16 KiB of chunks + up to
    // 128 KiB of ranks): the shapes that used to fall to the fp32 tile kernels at 8 waves per CU (depth <= 8) or to the generic kernel (deeper)
    Variant{"q16w_d8_c8_u4_gl_s2_cm_x", kKindQ16, 8, kQTile, 1, 8, 4, 1, 7 | 64, &launch_q16<8, 8, 4, 23 | 64>},
    Variant{"q16w_d8_c8_u4_gl", kKindQ16, 8, kQTile, 1, 8, 4, 1, 1 | 64, &launch_q16<8, 8, 4, 1 | 64>},   // (stream-order image: also the fp64 sum)
    // (depth 6 measured and NOT instantiated: 300 x d6 x 40 / 64 features, 10 M tuples -- 2164 / 1808 Mtuples/s against 2146 / 2085 on the fp32
    // tile kernel d6_t512_r1_c16_u8_dma: with 8 chains per lane the fp32 kernel holds its own at shallow depth; profiles/r05_wide_and_deep_ab.md)
    // depth 8 (BASELINE configs 3 and 5): tree = 3 KiB.  suffix _f = last level fused with its leaves.  Experiment variants
    // that no choice uses any more were removed in round 2 (register-staged chunks, R = 2, the unfused forms, 8-chain
    // stream kernels; their measurements stay in profiles/archive/r01_sweep_*.json)
    DDT_V("d8_t1024_r1_c4_u4_dma_f", 8, 1024, 1, 4, 4, 1, 1),
    DDT_V("d8_t512_r1_c8_u8_dma_f", 8, 512, 1, 8, 8, 1, 1),
    DDT_V("d8_t512_r1_c4_u4_dma_f", 8, 512, 1, 4, 4, 1, 1),  // 33..64 words per tuple: 128 KiB tile + 2 x 12 KiB chunks
    DDT_V("d8_t256_r1_c4_u4_dma", 8, 256, 1, 4, 4, 1, 0),
    DDT_V("d8_t128_r1_c8_u8_dma", 8, 128, 1, 8, 8, 1, 0),     // very wide tuples (up to ~220 words): 128-tuple tile, 8 chains per lane
    DDT_V("d8_t64_r1_c8_u8_dma", 8, 64, 1, 8, 8, 1, 0),       // up to ~440 words: one wave per CU, still 10x the generic kernel's global gathers
    // depth 6 (BASELINE config 2): tree = 768 B
    DDT_V("d6_t1024_r1_c16_u4_dma", 6, 1024, 1, 16, 4, 1, 0),
    DDT_V("d6_t512_r1_c16_u8_dma", 6, 512, 1, 16, 8, 1, 0),
    DDT_V("d6_t256_r1_c16_u4_dma", 6, 256, 1, 16, 4, 1, 0),
    DDT_V("d6_t128_r1_c16_u8_dma", 6, 128, 1, 16, 8, 1, 0),
    DDT_V("d6_t64_r1_c16_u8_dma", 6, 64, 1, 16, 8, 1, 0),
    // depth 4: tree = 192 B
    DDT_V("d4_t256_r1_c64_u8_dma", 4, 256, 1, 64, 8, 1, 0),
    DDT_V("d4_t128_r1_c64_u8_dma", 4, 128, 1, 64, 8, 1, 0),
    // depths 7, 5, 3: 12 KiB chunks like the depth-8 kernel
    DDT_V("d7_t1024_r1_c8_u4_dma", 7, 1024, 1, 8, 4, 1, 0),
    DDT_V("d7_t256_r1_c8_u4_dma", 7, 256, 1, 8, 4, 1, 0),
    DDT_V("d7_t128_r1_c8_u8_dma", 7, 128, 1, 8, 8, 1, 0),
    DDT_V("d5_t1024_r1_c32_u4_dma", 5, 1024, 1, 32, 4, 1, 0),
    DDT_V("d5_t256_r1_c32_u4_dma", 5, 256, 1, 32, 4, 1, 0),
    DDT_V("d5_t128_r1_c32_u8_dma", 5, 128, 1, 32, 8, 1, 0),
    DDT_V("d3_t256_r1_c128_u8_dma", 3, 256, 1, 128, 8, 1, 0),
    DDT_V("d3_t128_r1_c128_u8_dma", 3, 128, 1, 128, 8, 1, 0),
    // resident-model streaming kernels (small ensembles, HBM-bound; BASELINE config 1 is depth 4)
    DDT_S("stream_d4_u4_l4", 4, 4, 4),
    DDT_S("stream_d4_u4_l8", 4, 4, 8),
    DDT_S("stream_d6_u4_l4", 6, 4, 4),
    DDT_S("stream_d6_u4_l8", 6, 4, 8),
    DDT_S("stream_d8_u4_l8", 8, 4, 8),
    DDT_S("stream_d7_u4_l8", 7, 4, 8),
    DDT_S("stream_d5_u4_l8", 5, 4, 8),
    DDT_S("stream_d3_u4_l8", 3, 4, 8),
};

// ids: the perfect-tree kernels above, then the deep perfect-tree kernels of ddt_deep.hip, then the sparse-forest kernels of ddt_sparse.hip
static constexpr int kDenseVariants = (int)(sizeof(g_variants) / sizeof(g_variants[0]));
int num_variants() { return kDenseVariants + num_deep_variants() + num_sparse_variants() + num_sparse_r_variants(); }
const Variant& variant(int i) {
  if (i < kDenseVariants) return g_variants[i];
  i -= kDenseVariants;
  if (i < num_deep_variants()) return deep_variant(i);
  i -= num_deep_variants();
  return i < num_sparse_variants() ? sparse_variant(i) : sparse_r_variant(i - num_sparse_variants());
}

}  // namespace ddt
