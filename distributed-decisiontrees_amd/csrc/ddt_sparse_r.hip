// ddt_sparse_r.hip -- SPARSE forests on 32-bit ranks: pair records on EVERY level below the top image (round 6; BASELINE config 4).
//
// Why: the sparse kernel of ddt_sparse.hip is bound by the NUMBER of gather wave-instructions it issues (7.0 per tree and wave on
// 512 trees x depth <= 16 x 64 features, ~33 cycles of the CU's vector-memory pipe each = 89 % of its cycles, profiles/r05_pmc_cfg2_cfg4_cfg6.md
// section 7): one 16-byte record {fp32 key, w, left, right} decides ONE level.  Three nodes fit 16 bytes only as one-word nodes, i.e. with
// RANKS for thresholds -- and u16 ranks (the q16 pre-pass) stop at 65 k distinct thresholds per feature where this forest has ~100 k.  So the
// ranks here are 17 bits wide and the feature tile stays 32 bits per value (the fp32 tile's size: two blocks of 256 tuples x 64 features per CU,
// like `sparse_dp_k8_u8_t256`); a node is ONE word {rank : 17 | feature : 7 | flags : 8}, the tile holds rank(x) << 15 | 0x7FFF, and the compare
// !(x < t) (DTPU.sv:653-657) is ONE unsigned compare of the two words (ddt_internal.h "32-bit ranks").  Per tree:
//   top     K levels out of LDS, 4 bytes per node (K = 9 in the 16 KiB that held K = 8 as 8-byte records)
//   deep    16-byte PAIR records {node, left child, right child, ptr}: one gather decides TWO levels on every level below; early leaves
//           are the child words themselves (flag in the node), a leaf two levels down is a LEAF record in the grandchild's slot.
// Depth 16 with K = 9: 4 gather instructions per tree and wave.  The per-node work is the reference's (read node -> gather feature ->
// compare / missing rule -> next node, DTPU.sv:579-720), the sums run in its adder order (FPAddersReduceTree.sv:94-141, FPAggregator.v:79-131,
// Core.sv:486-541).  What the ranks cost: a pre-pass per batch (transpose + rank32_kernel below) that the fp32-tile kernels do not have.
// The kernel is latency-bound at two waves per SIMD (LDS: 80 KiB per block of four waves); what the walk does about it: four trees' visits advance
// together (half rounds), the next group's top levels are walked behind the deep rounds, and the late rounds of a group run under the early rounds
// of the next (sparse_r_walk_lag).
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdlib>

#include "ddt_device.h"
#include "ddt_internal.h"

namespace ddt {

// ---------------------------------------------------------------------------------------------------
// rank32_kernel: r(x) = #{keys <= x} against tables too long for LDS (100 k keys = 400 KiB per feature).  LDS holds the feature's DIRECTORY --
// the last key of every block of 2^blk_log2 keys, searched like rank_kernel's table (bucket lookup + log2 P probes, ddt_kernels.hip) --
// which names the one block that holds the answer; that block comes from global memory (L2-resident: the blocks of all features are a few
// tens of MB) with ONE 16-byte gather per four keys.  Output: x' = r << 15 | 0x7FFF in tiles [n_pad / T][W][T] (a missing value: 0xFFFFFFFF and
// its tile's flag), what score_sparse_r_kernel DMAs into LDS.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kR32Threads = 1024;

template <int POLICY>
__global__ __launch_bounds__(kR32Threads) void rank32_kernel(const uint32_t* __restrict__ xT, uint64_t n, uint64_t n_pad, const uint32_t* __restrict__ dir,
                                                             uint32_t Dpad, const uint32_t* __restrict__ tabP, const uint16_t* __restrict__ tabS,
                                                             const uint32_t* __restrict__ tab, uint32_t tab_bytes, uint32_t blk_log2, uint32_t miss_raw,
                                                             uint32_t ieee, uint32_t W, uint32_t tile_log2, uint32_t* __restrict__ r32,
                                                             uint32_t* __restrict__ tile_flags) {
  // Feature -> XCD affinity: workgroups go to the 8 XCDs round-robin by their linear id, and every value's key-block gather wants its feature's
  // blocks in THAT XCD's L2 (100 k keys = 400 KiB per feature, 25 MB for 64 features against 4 MB of L2 per XCD).  So the grid is linear and
  // feature j is only ever worked on by the blocks with id % 8 == j % 8: an XCD sees W / 8 features' tables.
  const uint32_t tid = threadIdx.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, fper = (W + 7u) / 8u;
  const uint32_t j = xcd + 8u * (slot % fper), chunk = slot / fper, chunks = gridDim.x / (8u * fper);
  if (j >= W) return;
  // (a linear table, the search position carried as the byte address of its entry: rank_kernel, ddt_prepass.hip)
  for (uint32_t i = tid; i < Dpad; i += kR32Threads) lds_st_u32(i * 4u, dir[(size_t)j * Dpad + i]);
  const uint32_t starts_off = Dpad * 4u;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(tabS + (size_t)j * kQ16RankBuckets);
    for (uint32_t i = tid; i < kQ16RankBuckets / 2u; i += kR32Threads) lds_st_u32(starts_off + i * 4u, src[i]);
  }
  __syncthreads();
  const uint32_t Kd = tabP[j * 8u + 0u], lo = tabP[j * 8u + 1u], shift = tabP[j * 8u + 3u], P = tabP[j * 8u + 4u];
  const uint32_t koff = tabP[j * 8u + 5u], K = tabP[j * 8u + 6u], hi_real = tabP[j * 8u + 7u];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(tab), 0, (int)tab_bytes, 0x00020000);
  const uint32_t quads = 1u << (blk_log2 - 2u);  // 16-byte gathers per block
  const uint32_t last = (Dpad - 1u) * 4u;        // byte address of the directory's last entry: always an INT_MAX pad
  const bool clamp = __builtin_amdgcn_readfirstlane((int)(Kd + P > Dpad)) != 0;  // a search probes entries <= start + P - 2: inside the pads otherwise
  constexpr int ILP = 4;  // independent searches per lane: the dependent LDS reads of one search are latency bound
  const uint64_t pass_rows = (uint64_t)kR32Threads * ILP, pass_stride = (uint64_t)chunks * pass_rows;
  auto load_pass = [&](uint32_t (&dst)[ILP], uint64_t row0) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      const uint64_t row = row0 + (uint64_t)i * kR32Threads + tid;
      dst[i] = row < n_pad ? __builtin_nontemporal_load(xT + (uint64_t)j * n_pad + row) : 0u;  // (streamed once: keep the L2 for the key blocks)
    }
  };
  auto probes = [&](auto clamp_tag, const int32_t (&x)[ILP], uint32_t (&pos)[ILP]) {  // as rank_kernel's (ddt_prepass.hip): compare + select + add per probe
    constexpr bool CL = decltype(clamp_tag)::value;
    for (uint32_t step = P >> 1; step >= 64u; step >>= 1) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        uint32_t a = pos[i] + (step - 1u) * 4u;
        if (CL) a = a < last ? a : last;
        if ((int32_t)lds_u32(a) <= x[i]) pos[i] += step * 4u;
      }
    }
#pragma unroll
    for (uint32_t step = 32u; step >= 1u; step >>= 1) {
      if (step < P) {  // wave-uniform
        const uint32_t bound = last - (step - 1u) * 4u;
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
          const uint32_t a = CL ? (pos[i] < bound ? pos[i] : bound) : pos[i];
          if ((int32_t)lds_u32(a + (step - 1u) * 4u) <= x[i]) pos[i] += step * 4u;
        }
      }
    }
  };
  uint32_t raw_next[ILP];
  load_pass(raw_next, (uint64_t)chunk * pass_rows);
  for (uint64_t row0 = (uint64_t)chunk * pass_rows; row0 < n_pad; row0 += pass_stride) {
    uint32_t raw[ILP], pos[ILP];
    int32_t x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) raw[i] = raw_next[i];
    load_pass(raw_next, row0 + pass_stride);  // the next pass's column values fly while this pass searches
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      x[i] = (int32_t)(ieee ? ieee_key(raw[i]) : raw[i]);
      uint32_t b = ((uint32_t)x[i] - lo) >> shift;  // wraps to a huge value below lo: selected away next
      b = b < kQ16RankBuckets - 1u ? b : kQ16RankBuckets - 1u;
      b = x[i] < (int32_t)lo ? 0u : b;
      pos[i] = 4u * (uint32_t)*reinterpret_cast<const DDT_LDS(uint16_t)*>(starts_off + b * 2u);  // byte address of the bucket's first entry
    }
    if (clamp) probes(std::true_type{}, x, pos);
    else probes(std::false_type{}, x, pos);
#pragma unroll
    for (int i = 0; i < ILP; ++i) pos[i] >>= 2;
    // pos = number of blocks whose LAST key is <= x: all their keys count, and block `pos` holds the rest of the answer
    uint32_t cnt[ILP], byte[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      pos[i] = pos[i] < Kd ? pos[i] : Kd;  // (block Kd is the all-pad block behind the feature's keys)
      byte[i] = (koff + (pos[i] << blk_log2)) * 4u;
      cnt[i] = 0u;
    }
    for (uint32_t qd = 0; qd < quads; ++qd) {
      u32x4 v[ILP];
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (POLICY == 0) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, byte[i] + 16u * qd, 0, 0);
        else if (POLICY == 1) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, byte[i] + 16u * qd, 0, 16);  // sc1: served by the L2, no L1 line
        else v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, byte[i] + 16u * qd, 0, 2);                      // nt
      }
#pragma unroll
      for (int i = 0; i < ILP; ++i)
        cnt[i] += ((int32_t)v[i].x <= x[i] ? 1u : 0u) + ((int32_t)v[i].y <= x[i] ? 1u : 0u) + ((int32_t)v[i].z <= x[i] ? 1u : 0u) +
                  ((int32_t)v[i].w <= x[i] ? 1u : 0u);
    }
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      const uint64_t row = row0 + (uint64_t)i * kR32Threads + tid;
      uint32_t r = (pos[i] << blk_log2) + cnt[i];
      r = r < K ? r : K;
      r = x[i] >= (int32_t)hi_real ? K : r;  // (also what keeps the INT_MAX pads of the last block out of the count)
      uint32_t out = (r << kSrRankShift) | ((1u << kSrRankShift) - 1u);
      if (raw[i] == miss_raw && row < n) {  // bit equality with the missing pattern (DTPU.sv:653), before any transform
        out = kSrMissing;
        atomicOr(&tile_flags[row >> tile_log2], 1u);
      }
      if (row < n_pad) __builtin_nontemporal_store(out, r32 + ((((row >> tile_log2) * W + j) << tile_log2) + (row & ((1u << tile_log2) - 1u))));
    }
  }
}

hipError_t launch_r32_prepass(const ScoreArgs& a, const SparseAux& x, hipStream_t s) {
  const Q16Aux& q = x.q16;
  const R32Aux& r = x.r32;
  const uint32_t W = a.tuple_words;
  uint32_t tile_log2 = 0;
  while ((1u << tile_log2) < r.tile) ++tile_log2;
  const uint64_t tiles = q.n_pad >> tile_log2;
  if (tiles == 0) return hipSuccess;
  hipError_t e = launch_zero_words(q.tile_flags, tiles, s);
  if (e != hipSuccess) return e;
  e = launch_transpose(a.tuples, W, a.n, q.n_pad, q.xT, s);
  if (e != hipSuccess) return e;
  const uint32_t lds = q.Kpad * 4u + kQ16RankBuckets * 2u;
  static const int policy = [] {  // A/B: cache policy of the key-block gathers (DDT_R32_POLICY = 0 default / 1 sc1 / 2 nt)
    const char* v = getenv("DDT_R32_POLICY");
    return v && v[0] ? atoi(v) : 0;
  }();
  auto kern = policy == 1 ? rank32_kernel<1> : policy == 2 ? rank32_kernel<2> : rank32_kernel<0>;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  // long-lived blocks: the directory (up to 128 KiB) is loaded once per block -- resident blocks per CU x CUs, split over the features
  const uint32_t per_cu = lds <= 80u * 1024u ? 2u : 1u;
  uint32_t bx = (uint32_t)((q.n_pad + kR32Threads * 4u - 1u) / (kR32Threads * 4u));  // row chunks per feature
  const uint32_t fper = (W + 7u) / 8u;                                                // features per XCD
  const uint32_t want = (per_cu * a.num_cus + 8u * fper - 1u) / (8u * fper);
  if (bx > want) bx = want < 1u ? 1u : want;
  hipLaunchKernelGGL(kern, dim3(8u * fper * bx), dim3(kR32Threads), lds, s, q.xT, a.n, q.n_pad, q.tables, q.Kpad, q.tabP, q.tabS, r.tab, r.tab_bytes,
                     r.blk_log2, a.miss_raw, a.ieee, W, tile_log2, r.r, q.tile_flags);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// The walk.  lane = tuple, U = 8 trees (one PU group) in lock-step, THREADS tuples per block with their rank words feature-major in LDS.
// ---------------------------------------------------------------------------------------------------
// Where the rank tile of a block lives in LDS (behind the top images of one pass) and how a lane finds a feature's row.
//   rows of THREADS tuples (WP = false): [feature][THREADS x 4 bytes] at FEAT_OFF = the top images rounded up to a row -- the pre-pass's own layout.
//   wave-private rows (WP = true, tuples of up to 64 words): [wave][feature][64 x 4 bytes]; a wave's region is `ws` = 256 bytes x the tuple words
//   rounded up to a power of two, the regions start at a multiple of ws, so `(node word & mask) | lane_off` IS the address (the feature number sits
//   at bit 8 of a node word: ddt_internal.h).  Same bank behaviour as before: the 64 lanes of a wave read 64 consecutive words.
template <int THREADS, int STEPB, bool WP>
struct SrTile {
  uint32_t feat_off, lane_off, mask, ws_log2;
  __device__ __forceinline__ SrTile(uint32_t W, int tid) {
    if constexpr (WP) {
      ws_log2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(40u - (uint32_t)__builtin_clz(W - 1u)));  // 256 * 2^ceil(log2 W); W >= 4
      const uint32_t ws = 1u << ws_log2;
      feat_off = ws > (uint32_t)STEPB ? ws : (uint32_t)STEPB;  // both powers of two: the larger is a multiple of the other
      lane_off = feat_off + (((uint32_t)tid >> 6) << ws_log2) + ((uint32_t)tid & 63u) * 4u;
      mask = (ws - 1u) & ~0xFFu;
    } else {
      constexpr uint32_t ROWB = (uint32_t)THREADS * 4u;
      ws_log2 = 0u;
      feat_off = (uint32_t)((STEPB + ROWB - 1) / ROWB * ROWB);
      lane_off = feat_off + (uint32_t)tid * 4u;
      mask = kSrFeatMask;
    }
  }
};

// x' >= rec, or the node's missing direction for a missing value (DTPU.sv:653-667).  Logical operators on purpose: hipcc keeps such lane
// predicates as SGPR masks (ddt_sparse.hip sp_right)
template <bool SLOW>
__device__ __forceinline__ bool sr_right(uint32_t f, uint32_t rec) {
  const bool ge = f >= rec;
  if (!SLOW) return ge;
  const bool miss = f == kSrMissing, mr = (rec & kSrMissRight) != 0u;
  return (miss && mr) || (!miss && ge);
}

// One PU group (U = 8 trees, lock-step): K levels over the one-word nodes of the top images in LDS (1-based heap in bytes: m4 <- 2 m4 + 4 right), then
// ROUNDS of pair records, one 16-byte gather per TWO levels: a rotating pipeline of U chains, gathers unconditional and in a fixed order
// (ddt_sparse.hip); a finished walker gathers from beyond the resource's range (zeros, no cache touched).
//   * A round = two HALF rounds of four trees whose visits advance together, stage by stage -- four feature reads of the nodes in flight, then four of
//     the children, then the four gathers back to back -- so that a half round exposes the LDS latency twice, not eight times (one visit after the
//     other: 29.3 ms against a floor of 17 on BASELINE config 4); the other half's gathers fly meanwhile.
//   * A leaf's VALUE goes through the child's compare as if it were a node word: its feature bits name some row of the tile (or, with rows of THREADS
//     tuples, of the LDS beyond the block's allocation, where a DS read returns 0), and whatever comes out is never used: the walker is done.
//   * Software-pipelined ACROSS PU groups: the deep rounds of group g are latency-bound at two waves per SIMD (a round = LDS read -> compare -> LDS
//     read -> compare -> gather, then ~1 us until the records are back), and the top walk of group g + 1 -- K levels of LDS reads and VALU work --
//     needs nothing of group g.  Its images are in LDS as soon as group g's top walk is over (the DMA is issued behind the barrier that ends it), so
//     a wave walks a few top levels of g + 1 behind every deep round of g: the gathers fly under work instead of under a wait (config 4: 336 vs 327
//     Mtuples/s, profiles/r06_sparse_r32.md).  Two barriers per group.  The barrier that publishes the next images waits with a COUNTED `vmcnt(U)`:
//     the DMA is older than exactly the U first gathers of the group that were issued behind it (operations return in order) --
//     tools/check_dma_waits.py proves it on the binary.  The order of the sums is untouched: group g is folded before group g + 1's deep phase begins.
//   * The last round is visits only: a walker that is still alive stands on a record whose taken side is a leaf (the host counted the rounds:
//     SparseAux::max_rounds); a finished one has the zeros its out-of-range gather returned -- no leaf flag.  A wave that leaves the rounds early
//     consumes its last (idle) gathers, so that both exits reach the next group with nothing outstanding in the compiler's books.
template <int K, int U, int THREADS, bool SLOW, bool WP>
__device__ __forceinline__ void sparse_r_walk(const ScoreArgs& a, const SparseAux& x, const int tid, RefAcc<1>& ra, double& dacc, const uint4* img, const uint32_t groups) {
  static_assert(U == 8 || U == 16, "the counted wait below is vmcnt(U)");
  constexpr int TOPB = 4 << K;
  constexpr int STEPB = U * TOPB;
  constexpr uint32_t ROWB = (uint32_t)THREADS * 4u;
  constexpr uint32_t ROW_LOG2 = THREADS == 512 ? 11u : THREADS == 256 ? 10u : 9u;
  static_assert((1u << ROW_LOG2) == ROWB, "tiles of 128 / 256 / 512 tuples");
  const SrTile<THREADS, STEPB, WP> tl(a.tuple_words, tid);
  const uint32_t lane_off = tl.lane_off, fmask = tl.mask;
  auto feat = [&](uint32_t rec) -> uint32_t {
    uint32_t addr;
    if constexpr (WP) asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(addr) : "v"(rec), "s"(fmask), "v"(lane_off));
    else asm("v_and_b32 %0, 0x7f00, %1\n\tv_lshl_add_u32 %0, %0, %2, %3" : "=&v"(addr) : "v"(rec), "n"(ROW_LOG2 - kSrFeatShift), "v"(lane_off));
    return lds_u32(addr);
  };
  const uint32_t C = a.clusters;
  const uint32_t n_steps = groups * 8u / (uint32_t)U;  // (`img`, `groups`: the launch's image, or a slice of it -- score_sparse_r_kernel)
  const uint32_t max_rounds = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.max_rounds);
  // top levels of the next group walked behind every deep round (the rest behind the last one)
  const uint32_t per_round = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)K + (max_rounds > 1u ? max_rounds - 1u : 1u) - 1u) / (max_rounds > 1u ? max_rounds - 1u : 1u)));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(x.deep), 0, (int)x.deep_bytes, 0x00020000);
  const uint32_t idle_off = x.idle_off;
  uint32_t m4[U];
  auto top_reset = [&]() {
#pragma unroll
    for (int u = 0; u < U; ++u) m4[u] = 4u;
  };
  auto top_levels = [&](uint32_t count) {  // `count` more levels of the top walk over the images in LDS (level-independent code: a runtime loop)
    for (uint32_t l = 0; l < count; ++l) {
      uint32_t nd[U], f[U];
#pragma unroll
      for (int u = 0; u < U; ++u) nd[u] = lds_u32(m4[u] + (uint32_t)(u * TOPB));
#pragma unroll
      for (int u = 0; u < U; ++u) f[u] = feat(nd[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) m4[u] = (m4[u] << 1) + (sr_right<SLOW>(f[u], nd[u]) ? 4u : 0u);
    }
  };
  u32x4 rr[U];
  // ---- prologue: group 0's top walk, the plain way ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // images of group 0 and the rank tile are in LDS for everyone
  top_reset();
  top_levels((uint32_t)K);
  {
    uint32_t cb[U];  // (read before the barrier: behind it the DMA of the next images may land)
#pragma unroll
    for (int u = 0; u < U; ++u) cb[u] = lds_u32((uint32_t)(u * TOPB));
    __syncthreads();
    if (1u < n_steps) dma_chunk<THREADS, STEPB>(img, 1, 0, tid);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, cb[u] + (m4[u] << 2), 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  for (uint32_t g = 0; g < n_steps; ++g) {
    const bool next = g + 1u < n_steps;
    if (next) {
      if constexpr (U == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the DMA of the next images is older than the U gathers issued behind it
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      __syncthreads();                                   // ... and has landed for every wave
      top_reset();
    }
    uint32_t lv = 0;  // levels of group g + 1 walked so far
    bool act[U];
    float leafv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      act[u] = true;
      leafv[u] = 0.f;
    }
    bool alive = true;
    if (max_rounds > 1u) {
      uint32_t r = 1u;
      do {
        bool any = false;
#pragma unroll
        for (int h = 0; h < U; h += 4) {
          uint32_t fn[4], fc[4], cw[4];
          bool r0[4], leaf[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            asm volatile("" : "+v"(rr[h + i].x), "+v"(rr[h + i].y), "+v"(rr[h + i].z), "+v"(rr[h + i].w));
            fn[i] = feat(rr[h + i].x);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            r0[i] = sr_right<SLOW>(fn[i], rr[h + i].x);
            cw[i] = r0[i] ? rr[h + i].z : rr[h + i].y;
            leaf[i] = (rr[h + i].x & (r0[i] ? kSrRightLeaf : kSrLeftLeaf)) != 0u;
            fc[i] = feat(cw[i]);  // (a leaf's value read as a node word: above)
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool r1 = sr_right<SLOW>(fc[i], cw[i]);
            const uint32_t nxt = rr[h + i].w + (r0[i] ? 32u : 0u) + (r1 ? 16u : 0u);
            if (act[h + i] && leaf[i]) leafv[h + i] = __uint_as_float(cw[i]);
            act[h + i] = act[h + i] && !leaf[i];
            any = any || act[h + i];
            rr[h + i] = __builtin_amdgcn_raw_buffer_load_b128(rs, act[h + i] ? nxt : idle_off, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        alive = __ballot(any) != 0ull;
        if (next) {  // the next group's top walk, a few levels behind every round: the gathers above fly meanwhile
          const uint32_t c = (uint32_t)K - lv < per_round ? (uint32_t)K - lv : per_round;
          top_levels(c);
          lv += c;
          __builtin_amdgcn_sched_barrier(0);
        }
      } while (alive && ++r < max_rounds);
    }
    if (!alive) {
#pragma unroll
      for (int u = 0; u < U; ++u) asm volatile("" : : "v"(rr[u].x), "v"(rr[u].y), "v"(rr[u].z), "v"(rr[u].w));
    }
    if (alive) {  // the last round: visits only
#pragma unroll
      for (int h = 0; h < U; h += 4) {
        uint32_t fn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          asm volatile("" : "+v"(rr[h + i].x), "+v"(rr[h + i].y), "+v"(rr[h + i].z), "+v"(rr[h + i].w));
          fn[i] = feat(rr[h + i].x);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool r0 = sr_right<SLOW>(fn[i], rr[h + i].x);
          if ((rr[h + i].x & (r0 ? kSrRightLeaf : kSrLeftLeaf)) != 0u) leafv[h + i] = __uint_as_float(r0 ? rr[h + i].z : rr[h + i].y);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (next && lv < (uint32_t)K) top_levels((uint32_t)K - lv);
#pragma unroll
    for (int h = 0; h < U / 8; ++h) {  // PU group by PU group, in stream order
      if (a.sum_mode == 1) {
#pragma unroll
        for (int u = 0; u < 8; ++u) dacc += (double)leafv[8 * h + u];
      } else {  // FPAddersReduceTree.sv:94-141, then the slot accumulate of this group's cluster
        const float lf[1][8] = {{leafv[8 * h + 0], leafv[8 * h + 1], leafv[8 * h + 2], leafv[8 * h + 3], leafv[8 * h + 4], leafv[8 * h + 5],
                                 leafv[8 * h + 6], leafv[8 * h + 7]}};
        double unused[1] = {0.0};
        fold_leaves<8, 1, 0>(lf, 0, C, ra, unused, a.sum_mode == 2);
      }
    }
    if (next) {
      uint32_t cb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) cb[u] = lds_u32((uint32_t)(u * TOPB));
      __syncthreads();  // every wave is through with the images of group g + 1
      if (g + 2u < n_steps) dma_chunk<THREADS, STEPB>(img, g + 2u, 0, tid);
#pragma unroll
      for (int u = 0; u < U; ++u) {  // ... exactly U gathers behind that DMA: what the counted wait at the loop's top counts
        rr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, cb[u] + (m4[u] << 2), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// The same walk with the deep rounds of CONSECUTIVE groups overlapped ("lag"): rounds 1 .. A of group g (A = half the rounds, rounded up) run
// interleaved with rounds A + 1 .. R of group g - 1 -- the late rounds, where a third of the walkers and less are still alive, no longer stand
// alone in front of their gathers' latency.  Two record sets per lane that swap roles from group to group (no register moves: a set's last gathers
// are in flight when the group changes); group g - 1 is folded inside group g's step, before group g: the order of the sums is untouched.
// Every iteration issues the same gathers in the same order (older set first, 2 x U, whatever is alive): the compiler's wait counts are exact on
// every path.  That needs walkers without an "active" mask: a finished walker is sent out of the buffer's range, gets {0, 0, 0, 0} = "feature 0 against
// rank 0, no leaf flag, next block at 0", lands in bytes 0..63 of the deep array -- records {0, 0, 0, 0xFFFFFFC0} (ddt_sparse_host.cpp) -- and from there
// goes out of range again: every second gather of a finished walker touches no cache, none needs a mask.
template <int K, int U, int THREADS, bool SLOW, bool WP, int NB, int ODD>
__device__ __forceinline__ void sparse_r_walk_lag(const ScoreArgs& a, const SparseAux& x, const int tid, RefAcc<1>& ra, double& dacc, const uint4* img, const uint32_t groups) {
  static_assert(U == 8, "the counted wait below is vmcnt(8)");
  constexpr int TOPB = 4 << K;
  constexpr int STEPB = U * TOPB;
  constexpr uint32_t ROWB = (uint32_t)THREADS * 4u;
  constexpr uint32_t ROW_LOG2 = THREADS == 512 ? 11u : THREADS == 256 ? 10u : 9u;
  static_assert((1u << ROW_LOG2) == ROWB, "tiles of 128 / 256 / 512 tuples");
  const SrTile<THREADS, STEPB, WP> tl(a.tuple_words, tid);
  const uint32_t lane_off = tl.lane_off, fmask = tl.mask;
  auto feat = [&](uint32_t rec) -> uint32_t {
    uint32_t addr;
    if constexpr (WP) asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(addr) : "v"(rec), "s"(fmask), "v"(lane_off));
    else asm("v_and_b32 %0, 0x7f00, %1\n\tv_lshl_add_u32 %0, %0, %2, %3" : "=&v"(addr) : "v"(rec), "n"(ROW_LOG2 - kSrFeatShift), "v"(lane_off));
    return lds_u32(addr);
  };
  const uint32_t C = a.clusters;
  const uint32_t n_steps = groups;
  // R = SparseAux::max_rounds = 2 NB + ODD rounds, compile-time: the iterations below are straight-line code (with a runtime trip count the
  // compiler's wait counts at the loop header were those of the loop's entry -- 7 where the back edge has 15 gathers behind the record)
  constexpr uint32_t B = (uint32_t)NB, A = (uint32_t)(NB + ODD);  // rounds of a group walked in the next group's step / in its own
  constexpr uint32_t per_round = ((uint32_t)K + A - 1u) / A;     // top levels of the next group behind every iteration
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(x.deep), 0, (int)x.deep_bytes, 0x00020000);
  uint32_t m4[U];
  auto top_reset = [&]() {
#pragma unroll
    for (int u = 0; u < U; ++u) m4[u] = 4u;
  };
  auto top_levels = [&](uint32_t count) {
#pragma unroll
    for (uint32_t l = 0; l < count; ++l) {
      uint32_t nd[U], f[U];
#pragma unroll
      for (int u = 0; u < U; ++u) nd[u] = lds_u32(m4[u] + (uint32_t)(u * TOPB));
#pragma unroll
      for (int u = 0; u < U; ++u) f[u] = feat(nd[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) m4[u] = (m4[u] << 1) + (sr_right<SLOW>(f[u], nd[u]) ? 4u : 0u);
    }
  };
  struct Set {
    u32x4 rr[U];
    float leafv[U];
  };
  // one round of a set: the visits of its U records (two levels each) and the next gathers (none in a set's last round: `last_tag`)
  auto round = [&](Set& S, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
#pragma unroll
    for (int h = 0; h < U; h += 4) {
      uint32_t fn[4], fc[4], cw[4];
      bool r0[4], leaf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("" : "+v"(S.rr[h + i].x), "+v"(S.rr[h + i].y), "+v"(S.rr[h + i].z), "+v"(S.rr[h + i].w));
        fn[i] = feat(S.rr[h + i].x);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        r0[i] = sr_right<SLOW>(fn[i], S.rr[h + i].x);
        cw[i] = r0[i] ? S.rr[h + i].z : S.rr[h + i].y;
        leaf[i] = (S.rr[h + i].x & (r0[i] ? kSrRightLeaf : kSrLeftLeaf)) != 0u;
        if constexpr (!LAST) fc[i] = feat(cw[i]);  // (a leaf's value read as a node word: above)
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        S.leafv[h + i] = leaf[i] ? __uint_as_float(cw[i]) : S.leafv[h + i];
        if constexpr (!LAST) {
          const bool r1 = sr_right<SLOW>(fc[i], cw[i]);
          uint32_t nxt = S.rr[h + i].w + (r0[i] ? 32u : 0u) + (r1 ? 16u : 0u);
          asm volatile("" : "+v"(nxt));  // (computed for every lane: hipcc otherwise sinks the child's compare into a divergent branch on `leaf`)
          S.rr[h + i] = __builtin_amdgcn_raw_buffer_load_b128(rs, leaf[i] ? 0xFFFFFFF0u : nxt, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto fold = [&](Set& S) {  // FPAddersReduceTree.sv:94-141, then the slot accumulate of this group's cluster
    if (a.sum_mode == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) dacc += (double)S.leafv[u];
    } else {
      const float lf[1][8] = {{S.leafv[0], S.leafv[1], S.leafv[2], S.leafv[3], S.leafv[4], S.leafv[5], S.leafv[6], S.leafv[7]}};
      double unused[1] = {0.0};
      fold_leaves<8, 1, 0>(lf, 0, C, ra, unused, a.sum_mode == 2);
    }
  };
  auto first_gathers = [&](Set& S, const uint32_t (&cb)[U]) {  // the dense level-K records of a group (issued in front of the DMA of the images after next)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      S.rr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, cb[u] + (m4[u] << 2), 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      S.leafv[u] = 0.f;
    }
  };
  auto next_top = [&](uint32_t& lv) {  // the next group's top walk, a few levels behind every iteration: the gathers fly meanwhile
    const uint32_t c = (uint32_t)K - lv < per_round ? (uint32_t)K - lv : per_round;
#pragma unroll
    for (uint32_t l = 0; l < per_round; ++l)
      if (l < c) top_levels(1u);
    lv += c;
    __builtin_amdgcn_sched_barrier(0);
  };
  // The barrier that publishes the images of group g + 1 stands in front of that group's first top levels, not at the step's top: the rounds before it
  // read the rank tile only, so the DMA's flight (issued at the end of the step before) is covered by them.  Its wait is counted: the DMA is older than
  // the gathers of the rounds in between -- NWAIT, a constant of (NB, ODD); tools/check_dma_waits.py proves it.  (The first gathers of a group go out
  // IN FRONT of that DMA: operations return in order, and behind it they would come back only once its 16 KiB have landed.)
  constexpr uint32_t NWAIT = B >= 1u ? (uint32_t)U + (A + 1u < (uint32_t)(2 * NB + ODD) ? (uint32_t)U : 0u) : 0u;
  static_assert(NWAIT == 0u || NWAIT == 8u || NWAIT == 16u, "vmcnt immediate");
  auto publish_next = [&]() {
    if constexpr (NWAIT == 0u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (NWAIT == 8u) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();
    top_reset();
  };
  constexpr uint32_t RR = (uint32_t)(2 * NB + ODD);
  // group g's step: cur = its walkers (first gathers in flight, the youngest), prev = group g - 1's, A rounds in (at g = 0: idle walkers).
  // (`next` = there is a group g + 1, compile-time: with a runtime flag the paths with and without its first gathers meet at the loop's back edge, and
  // the compiler's wait counts for the other set's records become those of the path without -- 7 instead of 15)
  auto step = [&](Set& cur, Set& prev, const uint32_t g, auto next_tag) {
    constexpr bool next = decltype(next_tag)::value;
    uint32_t lv = 0;  // levels of group g + 1 walked so far
    auto iter = [&](auto i_tag) {  // iteration i: round A + 1 + i of the older set, round 1 + i of this group's
      constexpr uint32_t i = decltype(i_tag)::value;
      round(prev, std::bool_constant<A + 1u + i == RR>{});
      round(cur, std::bool_constant<1u + i == RR>{});
      if constexpr (next) {
        if constexpr (i == 0u) publish_next();
        next_top(lv);
      }
    };
    if constexpr (B >= 1u) iter(std::integral_constant<uint32_t, 0>{});
    if constexpr (B >= 2u) iter(std::integral_constant<uint32_t, 1>{});
    if constexpr (B >= 3u) iter(std::integral_constant<uint32_t, 2>{});
    static_assert(B <= 3u, "up to six rounds");
    if (g > 0u) fold(prev);  // group g - 1 is through its R rounds: folded in stream order
    if constexpr (A > B) {
      round(cur, std::bool_constant<A == RR>{});
      if constexpr (next) {
        if constexpr (B == 0u) publish_next();
        next_top(lv);
      }
    }
    if constexpr (next) {
#pragma unroll
      for (uint32_t l = 0; l < (uint32_t)K; ++l)
        if (l >= lv) top_levels(1u);
      uint32_t cb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) cb[u] = lds_u32((uint32_t)(u * TOPB));
      first_gathers(prev, cb);  // the set of group g - 1 is free: it becomes group g + 1's
      __syncthreads();          // every wave is through with the images of group g + 1
      if (g + 2u < n_steps) dma_chunk<THREADS, STEPB>(img, g + 2u, 0, tid);
    }
  };
  auto drain = [&](Set& S) {  // the last group's rounds A + 1 .. R
    if constexpr (B >= 1u) round(S, std::bool_constant<A + 1u == RR>{});
    if constexpr (B >= 2u) round(S, std::bool_constant<A + 2u == RR>{});
    if constexpr (B >= 3u) round(S, std::bool_constant<A + 3u == RR>{});
    fold(S);
  };
  Set S0, S1;
  // ---- prologue: group 0's top walk, the plain way ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // images of group 0 and the rank tile are in LDS for everyone
  top_reset();
  top_levels((uint32_t)K);
  {
    uint32_t cb[U];  // (read before the barrier: behind it the DMA of the next images may land)
#pragma unroll
    for (int u = 0; u < U; ++u) cb[u] = lds_u32((uint32_t)(u * TOPB));
#pragma unroll
    for (int u = 0; u < U; ++u) {  // the other set starts as eight finished walkers
      S1.rr[u] = u32x4{0u, 0u, 0u, 0u};
      S1.leafv[u] = 0.f;
    }
    first_gathers(S0, cb);
    __syncthreads();
    if (1u < n_steps) dma_chunk<THREADS, STEPB>(img, 1, 0, tid);
  }
  uint32_t g = 0;
  for (; g + 2u < n_steps; g += 2u) {
    step(S0, S1, g, std::true_type{});
    step(S1, S0, g + 1u, std::true_type{});
  }
  if (g + 1u < n_steps) {  // two groups left
    step(S0, S1, g, std::true_type{});
    step(S1, S0, g + 1u, std::false_type{});
    drain(S1);
  } else {  // one
    step(S0, S1, g, std::false_type{});
    drain(S0);
  }
}

// SPLIT: a second instantiation (of the walk without the lag only: a batch of a few tiles is not where the lag pays) -- with the slices as run-time
// selects in the one kernel, config 4's scoring took 23.2 instead of 23.03 ms (same box, alternating libraries: profiles/r06_raw/s45_*)
template <int K, int U, int THREADS, bool WP, int NB, int ODD, bool SPLIT = false>  // NB + ODD > 0: the walk with 2 NB + ODD rounds, consecutive groups overlapped
__global__ __launch_bounds__(THREADS) void score_sparse_r_kernel(const ScoreArgs a, const SparseAux x) {
  constexpr int TOPB = 4 << K;
  constexpr int STEPB = U * TOPB;
  constexpr int ROW = THREADS * 4;
  static_assert(U == 8 || U == 16, "one or two PU groups per pass");
  static_assert((STEPB / 16) % 64 == 0, "whole waves per DMA");
  const int tid = threadIdx.x;
  const uint64_t tile0 = (uint64_t)blockIdx.x * THREADS;
  const uint32_t W = a.tuple_words;

  // A batch of a few tiles (Q16Aux::split, sparse_launch): blockIdx.y = a slice of C consecutive PU groups starting at a multiple of C -- in an image in
  // stream order group g belongs to cluster g mod C (Core.sv:291-316), so within such a slice every cluster's accumulator takes exactly ONE group's
  // sum (x + 0): the walk is the uncut launch's, only the epilogue lets the ring out instead of its total, and launch_cm_combine runs the adds.
  const uint32_t g0 = SPLIT ? blockIdx.y * a.clusters : 0u;
  const uint4* img = SPLIT ? a.img + (size_t)g0 * (STEPB / 16) : a.img;
  const uint32_t groups = SPLIT ? (x.n_groups - g0 < a.clusters ? x.n_groups - g0 : a.clusters) : x.n_groups;
  dma_chunk<THREADS, STEPB>(img, 0, 0, tid);  // top images of the first pass
  {
    // the rank tile is one contiguous block of W * ROW bytes of the pre-pass's output, [feature][THREADS tuples]: DMA it in; the first barrier of the
    // walk publishes it.  A wave instruction fills 1 KiB of LDS with 64 pieces of 16 bytes from anywhere: for wave-private rows (SrTile) those are
    // four rows of one wave's region = the 256-byte pieces [64 w, 64 w + 64) of four features' rows
    const SrTile<THREADS, STEPB, WP> tl(W, tid);
    const uint4* src = reinterpret_cast<const uint4*>(x.r32.r + (uint64_t)blockIdx.x * W * (uint32_t)THREADS);
    const uint32_t units = WP ? (uint32_t)(THREADS / 64) << (tl.ws_log2 - 4u) : W * (ROW / 16);
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    for (uint32_t u0 = 0; u0 < units; u0 += THREADS) {
      const uint32_t lds_addr = tl.feat_off + (u0 + (uint32_t)wave_base) * 16u;
      const uint4* g = src + (u0 + (uint32_t)tid);
      bool valid = u0 + (uint32_t)wave_base < units;
      if constexpr (WP) {
        const uint32_t byte = (u0 + (uint32_t)tid) * 16u, w = byte >> tl.ws_log2, rem = byte & ((1u << tl.ws_log2) - 1u), f = rem >> 8;
        g = src + (f * (uint32_t)(ROW / 16) + w * 16u + ((rem & 255u) >> 4));
        valid = valid && (uint32_t)__builtin_amdgcn_readfirstlane((int)f) < W;  // (rows come in fours and W is a multiple of 4: a wave's unit is whole)
      }
      if (valid)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds_addr), "v"(g) : "memory");
    }
  }
  const bool slow = __builtin_amdgcn_readfirstlane((int)x.q16.tile_flags[blockIdx.x]) != 0;  // the tile holds a missing value

  RefAcc<1> ra;
  ra.init();
  double dacc = 0.0;
  const uint32_t C = a.clusters;
  if constexpr (NB + ODD > 0) {
    if (!slow) sparse_r_walk_lag<K, U, THREADS, false, WP, NB, ODD>(a, x, tid, ra, dacc, img, groups);
    else sparse_r_walk_lag<K, U, THREADS, true, WP, NB, ODD>(a, x, tid, ra, dacc, img, groups);
  } else {
    if (!slow) sparse_r_walk<K, U, THREADS, false, WP>(a, x, tid, ra, dacc, img, groups);
    else sparse_r_walk<K, U, THREADS, true, WP>(a, x, tid, ra, dacc, img, groups);
  }
  ra.align(C);
  const uint64_t row = tile0 + (uint64_t)tid;
  if constexpr (SPLIT) {  // out = [groups rounded up to C][n_pad] partial sums: cluster k's accumulator = the sum of group g0 + k (+ 0; nothing beyond the forest's last group)
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if ((uint32_t)k < C) a.out[(uint64_t)(g0 + (uint32_t)k) * x.q16.n_pad + row] = ra.a[0][k];
    return;
  }
  if (row < a.n) a.out[row] = (a.sum_mode == 1) ? (float)dacc : ra.total(0, C, a.sum_mode == 2);
}

template <int K, int U, int THREADS>
static hipError_t launch_sparse_r_v(const ScoreArgs& a, const Variant& v, hipStream_t s) {
  const SparseAux& x = *reinterpret_cast<const SparseAux*>(a.aux);
  const uint32_t lds = v.lds_bytes_sparse(a.tuple_words);
  // wave-private rows where their padding costs no block per CU (Variant::wave_rows: the same rule sized `lds`)
  // Consecutive groups overlapped (sparse_r_walk_lag) where the forest's longest path takes 2 .. 6 rounds and the LDS allows two blocks per CU at most:
  // that walk holds two record sets per lane (~180 VGPRs = two waves per SIMD; 128 x d14 x 20 features, three blocks per CU: 1335 vs 1578 Mtuples/s)
  static const bool lag_on = [] {  // A/B: DDT_SPARSE_R_LAG=0 -> a group's rounds alone, one group after the other
    const char* v = getenv("DDT_SPARSE_R_LAG");
    return !(v && v[0] == '0');
  }();
  const bool wp = v.wave_rows(a.tuple_words);
  const uint32_t rounds = (lag_on && 3u * lds > 160u * 1024u) ? x.max_rounds : 0u;
#define DDT_SR_KERN(NB, ODD) (wp ? score_sparse_r_kernel<K, U, THREADS, true, NB, ODD> : score_sparse_r_kernel<K, U, THREADS, false, NB, ODD>)
  auto kern = rounds == 2u ? DDT_SR_KERN(1, 0) : rounds == 3u ? DDT_SR_KERN(1, 1) : rounds == 4u ? DDT_SR_KERN(2, 0) : rounds == 5u ? DDT_SR_KERN(2, 1)
            : rounds == 6u ? DDT_SR_KERN(3, 0) : DDT_SR_KERN(0, 0);
#undef DDT_SR_KERN
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const uint64_t blocks = (a.n + THREADS - 1) / THREADS;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
  if (!x.q16.skip_prepass) {  // ranks + per-tile missing flags of this batch (reused by the other classes' launches)
    e = launch_r32_prepass(a, x, s);
    if (e != hipSuccess) return e;
  }
  if (a.ev_mid) (void)hipEventRecord(a.ev_mid, s);
  if (x.q16.split) {  // a batch of a few tiles: slices of C groups (sparse_launch)
    const uint32_t C = a.clusters ? a.clusters : 1u;
    auto ks = wp ? score_sparse_r_kernel<K, U, THREADS, true, 0, 0, true> : score_sparse_r_kernel<K, U, THREADS, false, 0, 0, true>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ks, dim3((uint32_t)blocks, (x.n_groups + C - 1u) / C), dim3(THREADS), lds, s, a, x);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(THREADS), lds, s, a, x);
  return hipGetLastError();
}

#define DDT_SPR(K, U, T) /* levels = K, threads = tile = T, opt bit 5 */ \
  Variant { "sparse_r_k" #K "_u" #U "_t" #T, kKindSparse, K, T, 1, U, U, 1, 32, &launch_sparse_r_v<K, U, T> }

static const Variant g_sparse_r_variants[] = {
    // two blocks of 256 tuples per CU up to 64 tuple words with K = 9 (2 x (16 + 64) KiB), K = 10 up to 48; 128 tuples per block beyond 64 words (K = 9: 16 +
    // 64 KiB at 128 words).  (K = 8 forms existed for 65..72 words and never won elsewhere: removed with the six-fold instantiation per round count)
    DDT_SPR(9, 8, 256), DDT_SPR(10, 8, 256), DDT_SPR(9, 8, 128),
    // (two PU groups per pass -- 16 chains per lane, K = 8 in the same 16 KiB -- measured and NOT instantiated: config 4 301 vs 335 Mtuples/s on
    // k9_u8: five rounds instead of four, profiles/EXPERIMENTS.md round 6; the walks above take U = 16 should it be wanted again)
};

int num_sparse_r_variants() { return (int)(sizeof(g_sparse_r_variants) / sizeof(g_sparse_r_variants[0])); }
const Variant& sparse_r_variant(int i) { return g_sparse_r_variants[i]; }

}  // namespace ddt
