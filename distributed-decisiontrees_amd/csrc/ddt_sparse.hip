// ddt_sparse.hip -- scoring kernel for SPARSE (explicit-children) forests: BASELINE config 4, deep random forests.
//
// What it replaces: nothing that works in the reference -- its engine only takes perfect trees that fit a PU's BRAM
// (rtl/DTEngine/core/DTPU.sv:20-28); the hook for bigger trees is the disabled hybrid path (entry bit 14 "next node
// is a leaf", DTPU.sv:637,661,675,712-715; PartialTrees, Core.sv:380 bit 8, DTPU.sv:736-745).  The per-node work is
// the reference's: read node -> gather feature -> compare / missing rule -> next node (DTPU.sv:579-720), leaf sums in
// the reference's adder order (FPAddersReduceTree.sv:94-141, FPAggregator.v:79-131, Core.sv:486-541).
//
// Mapping: lane = tuple, 256 tuples per block with their features feature-major in LDS (the per-lane gather x[fidx]
// is then bank-conflict free for any fidx), the 8 trees of a PU group walked in lock-step (8 independent dependent-
// load chains per lane).  A tree is walked in two phases:
//   top    the first K levels, stored as a perfect heap and staged in LDS by global->LDS DMA (one image per group,
//          single buffer: the DMA of group g+1 is issued right after the last LDS read of group g and overlaps the
//          deep phase of group g)
//   deep   16-byte records {thr, w, left, right} gathered from L2 / HBM: child pointers or leaf values are IN the
//          parent's record, so a visit is ONE 16-byte load and the leaf needs no extra access.  The 8 walks of a lane form a
//          ROTATING pipeline: the gather of tree u's next record is issued right after ITS visit and flies while the other
//          seven trees are visited.  Lanes that reached a leaf idle until the wave's deepest lane is done (__ballot early
//          exit, per wave).
//   "sparse_dk_*" (dense level K, Variant::opt bit 1): ALL K top levels are 8-byte records in LDS and level K is a dense block of
//          2^K deep records per tree addressed by the heap index (ddt_internal.h): a third less LDS per tree, so two K = 8 blocks
//          of 256 tuples x 64 features share a CU (one's top phase overlaps the other's deep phase) or one block of 512 holds K = 9.
// No MFMA: compare + gather.  Bound by the vector-memory gather rate of the deep phase.
#include <hip/hip_runtime.h>

#include "ddt_device.h"
#include "ddt_internal.h"

namespace ddt {

// Compare rule on a sparse record (thr, w): DTPU.sv:653-667, the missing direction in kSpMissRight.  Written with
// logical operators on purpose: hipcc keeps such lane predicates as SGPR masks (s_and / s_or), whereas a ternary
// between two predicates is lowered to 0/1 VGPRs and four extra VALU instructions per visit.
// Q (rank-quantised kernels): f is the u16 rank of the feature value, thr the node's rank R; x >= t  <=>  rank(x) >= R
// (ddt_prepass.hip "The rank-quantised path"), a missing value has the rank kQMissing.
template <bool SLOW, bool Q>
__device__ __forceinline__ bool sp_right(uint32_t f, uint32_t thr, uint32_t w, uint32_t miss_key) {
  const bool ge = Q ? f >= thr : (int32_t)f >= (int32_t)thr;
  if (!SLOW) return ge;
  const bool miss = f == (Q ? kQMissing : miss_key), mr = (int32_t)(w << 2) < 0;  // kSpMissRight = bit 29
  return (miss && mr) || (!miss && ge);
}

// GF ("sparse_gf_*", tuples too wide for a feature tile in LDS): the feature comes from the tuple's own row in global memory --
// gsrc = buffer resource over the block's rows, lane_off = byte offset of the lane's row in it, the record's address field = byte offset
// of the feature in the row -- and gets here the raw -> key transform the staged tiles get once (stage_word)
struct GfSrc {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t miss_raw, ieee;
};

template <bool Q, bool GF>
__device__ __forceinline__ uint32_t sp_feature(uint32_t w, uint32_t lane_off, const GfSrc& gs) {
  if constexpr (GF) {
    uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(gs.rsrc, (w & kSpAddrMask) + lane_off, 0, 0);
    if (gs.ieee) v = (v == gs.miss_raw) ? kMissSentinelIeee : ieee_key(v);  // wave-uniform branch
    return v;
  } else {
    const uint32_t addr = (w & kSpAddrMask) | lane_off;
    if (Q) return *reinterpret_cast<const DDT_LDS(uint16_t)*>(addr);  // ds_read_u16
    return lds_u32(addr);
  }
}

// The walk of all PU groups for one tile.  SLOW = the tile holds a missing value: apply the per-node missing rule
// (block-uniform choice, like the perfect-tree kernels).
// M ("sparse_dm<M>_*", dense MID levels, round 5; only with DK): the levels K .. K+M-1 are dense too, as 8-BYTE records {thr, w} that continue
// the heap in global memory (record of heap node h at byte cbase + 8 h; no child words: children by index, early leaves padded with dummies
// like the top image), and the dense block of 16-byte records is level K+M (byte cbase - 8 * 2^(K+M) + 16 h).  Why: the counters of round 5
// (profiles/r05_pmc_cfg2_cfg4_cfg6.md) put the kernel's bound at the L2's service of L1 misses -- 1.03 misses per tree and tuple, most of
// them on the levels right below the top image, whose 16-byte records (96 KiB per PU group for levels 8-9) overflow a CU's 32 KiB L1.  Half
// the bytes per hot record = twice the records per line and per L1.
template <int K, int U, int THREADS, bool SLOW, bool Q, bool DK, bool GF = false, int M = 0, bool PR = false>
__device__ __forceinline__ void sparse_walk(const ScoreArgs& a, const SparseAux& x, const int tid, RefAcc<1>& ra, double& dacc, const GfSrc& gs) {
  static_assert(M == 0 || DK, "dense mid levels extend the dense level K");
  static_assert(!PR || (DK && M == 0 && !GF), "pair records: a feature tile in LDS, dense levels K and K + 1");
  constexpr int TOPB = (DK ? 8 : 12) << K;
  constexpr int STEPB = U * TOPB;  // U trees per pass: one PU group (or two)
  // Q: the u16 tile of the q16 pre-pass -- tuples t and t + 512 of a tile share a dword (rank_kernel)
  const uint32_t lane_off = GF ? (uint32_t)tid * a.tuple_words * 4u : Q ? ((((uint32_t)tid & 511u) << 2) | (((uint32_t)tid >> 9) << 1)) : (uint32_t)tid * 4u;
  const uint32_t miss_key = a.miss_key, C = a.clusters;
  const uint4* __restrict__ deep = x.deep;
  const uint32_t n_steps = x.n_groups * 8u / (uint32_t)U;  // the host pads the image to whole passes
  const uint32_t max_rounds = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.max_rounds);
  for (uint32_t g = 0; g < n_steps; ++g) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the top images of this pass (and, first pass, the feature tile) are in LDS for everyone

    // ---- top phase: levels 0..K-2 over 8-byte heap records, level K-1 over 16-byte records (dense level K: all K levels over
    //      8-byte records, the level-K record's byte offset in the deep array follows from the heap index) ----
    uint32_t m8[U];
#pragma unroll
    for (int u = 0; u < U; ++u) m8[u] = 8u;
#pragma unroll
    for (int lvl = 0; lvl < K - 1; ++lvl) {
      uint2 nd[U];
#pragma unroll
      for (int u = 0; u < U; ++u) nd[u] = lds_u2(m8[u] + (uint32_t)(u * TOPB));
      uint32_t f[U];
#pragma unroll
      for (int u = 0; u < U; ++u) f[u] = sp_feature<Q, GF>(nd[u].y, lane_off, gs);
#pragma unroll
      for (int u = 0; u < U; ++u) m8[u] = (m8[u] << 1) + (sp_right<SLOW, Q>(f[u], nd[u].x, nd[u].y, miss_key) ? 8u : 0u);
    }
    uint4 r[U];  // m8 = 8 * heap index in [2^(K-1), 2^K): record at 4*2^K + 16*(m - 2^(K-1)) = 2*m8 - 4*2^K
    // dense level K: the level K-1 record is 8 bytes {key, w}; its children are deep records at byte 2 * (8 * child heap index) +
    // cbase, cbase = word 0 of the tree's image.  Like the 16-byte record it is visited in the first round of the deep phase.
    uint2 r8[U];
    uint32_t cb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (DK) {
        r8[u] = lds_u2(m8[u] + (uint32_t)(u * TOPB));
        if (M == 0 && !PR) cb[u] = lds_u32((uint32_t)(u * TOPB)) + (m8[u] << 2);  // byte offset of the LEFT child's record
        else cb[u] = lds_u32((uint32_t)(u * TOPB));                        // dense mid levels: cbase itself (8-byte records at cbase + 8 h)
      } else {
        r[u] = lds_u4((m8[u] << 1) - (uint32_t)(4 << K) + (uint32_t)(u * TOPB));
      }
    }
    __syncthreads();  // every wave holds its level K-1 records: the top image buffer is free
    if (g + 1 < n_steps) dma_chunk<THREADS, STEPB>(a.img, g + 1, 0, tid);  // overlaps the deep phase below

    // ---- deep phase: one 16-byte gather per visit; lanes whose tree has reached its leaf are masked off.  A rotating pipeline of U
    //      chains: the gather of tree u's next record is issued right after ITS visit, in a fixed order and UNCONDITIONALLY (a
    //      finished lane re-reads record 0: one shared line), so that hipcc counts the loads and every visit waits with vmcnt(U-1)
    //      for the oldest one only.  (Round 2's form -- all U visits, then all U gathers back to back -- left the queue empty
    //      during the visits: 212 vs 237 Mtuples/s on BASELINE config 4, profiles/archive/r03_sparse_schedules_and_blocks.json.)
    //      The gathers go through a buffer resource: 32-bit byte offsets (the host keeps the deep array below 2^28 records), no
    //      64-bit address arithmetic on the VALU.
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(deep), 0, (int)x.deep_bytes, 0x00020000);
    // Round 5: a finished lane's gather goes BEYOND the resource's range: the range check answers it with zeros and nothing reaches the L1.
    // Until then it re-read record 0 -- "one shared line", but the texture cache merges lanes only within small groups, and the counters
    // showed 11.3 G cache accesses per 4 M tuples for 8.3 G live lane visits (profiles/r05_pmc_cfg2_cfg4_cfg6.md): a quarter of the
    // kernel's accesses were finished lanes.  (Its LDS feature read then uses row 0 of the tile: in range, unused.)
    const uint32_t idle_off = x.idle_off;
    bool act[U];  // per lane: tree still walking (kept as lane masks in SGPRs, not as bits of a VGPR)
    float leafv[U];
    u32x4 rr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      act[u] = true;
      leafv[u] = 0.f;
      if (!DK) rr[u] = u32x4{r[u].x, r[u].y, r[u].z, r[u].w};
    }
    if constexpr (DK && M == 0 && !PR) {  // first round: level K-1 out of the registers, every walker goes on to its level-K record
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t f = sp_feature<Q, GF>(r8[u].y, lane_off, gs);
        const bool right = sp_right<SLOW, Q>(f, r8[u].x, r8[u].y, miss_key);
        rr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, cb[u] + (right ? 16u : 0u), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (DK && M > 0) {
      // dense mid levels: M rounds in which every walker is alive (no leaf flags: early leaves are padded), 8 gathers of 8 bytes in
      // flight per lane and round, the same rotation as below; the last round fetches the 16-byte record of level K+M
      u32x2 d8[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t f = sp_feature<Q, GF>(r8[u].y, lane_off, gs);
        const bool right = sp_right<SLOW, Q>(f, r8[u].x, r8[u].y, miss_key);
        m8[u] = (m8[u] << 1) + (right ? 8u : 0u);  // 8 * heap index at level K
        d8[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, cb[u] + m8[u], 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 1; j <= M; ++j) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          asm volatile("" : "+v"(d8[u].x), "+v"(d8[u].y));
          const uint32_t f = sp_feature<Q, GF>(d8[u].y, lane_off, gs);
          const bool right = sp_right<SLOW, Q>(f, d8[u].x, d8[u].y, miss_key);
          m8[u] = (m8[u] << 1) + (right ? 8u : 0u);
          if (j < M) d8[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, cb[u] + m8[u], 0, 0);
          else rr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, cb[u] + (m8[u] << 1) - (uint32_t)(8u << (K + M)), 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (PR) {
      // "sparse_dp_*" (dense PAIR records, round 5): the levels K and K + 1 of a tree are one block of 2^K 16-byte records {key of the level-K
      // node, keys of its two children, feature numbers of the three as bytes + their missing directions}: ONE gather decides two levels (the
      // dense-mid form takes one per level), the next record is the walker's node in the dense block of level K + 2, by heap index.  The
      // feature number, not its LDS address, is in the record (three of them share a word): address = tile base + number * ROW.
      constexpr uint32_t ROWB = (uint32_t)THREADS * (Q ? 2u : 4u);  // (rank-quantised: the u16 tile of the q16 pre-pass, keys = ranks)
      const uint32_t fbase = (uint32_t)(((U * TOPB + ROWB - 1) / ROWB) * ROWB) + lane_off;  // FEAT_OFF + the lane's column
      const uint32_t miss_f = Q ? kQMissing : miss_key;
      auto feat = [&](uint32_t j) -> uint32_t {
        const uint32_t addr = (j * ROWB) + fbase;
        if (Q) return *reinterpret_cast<const DDT_LDS(uint16_t)*>(addr);
        return lds_u32(addr);
      };
      auto ge = [&](uint32_t f, uint32_t key) -> bool { return Q ? f >= key : (int32_t)f >= (int32_t)key; };
      u32x4 pr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t f = sp_feature<Q, GF>(r8[u].y, lane_off, gs);
        const bool right = sp_right<SLOW, Q>(f, r8[u].x, r8[u].y, miss_key);
        m8[u] = (m8[u] << 1) + (right ? 8u : 0u);  // 8 * heap index at level K
        pr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, cb[u] + (m8[u] << 1), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        asm volatile("" : "+v"(pr[u].x), "+v"(pr[u].y), "+v"(pr[u].z), "+v"(pr[u].w));
        const uint32_t w = pr[u].w;
        const uint32_t fa = feat(w & 0xFFu);
        bool r0 = ge(fa, pr[u].x);
        if (SLOW) r0 = (fa == miss_f) ? ((w >> 24) & 1u) != 0u : r0;
        const uint32_t kc = r0 ? pr[u].z : pr[u].y;
        const uint32_t jc = (r0 ? (w >> 16) : (w >> 8)) & 0xFFu;
        const uint32_t fc = feat(jc);
        bool r1 = ge(fc, kc);
        if (SLOW) r1 = (fc == miss_f) ? ((w >> (r0 ? 26 : 25)) & 1u) != 0u : r1;
        m8[u] = (m8[u] << 2) + (r0 ? 16u : 0u) + (r1 ? 8u : 0u);  // 8 * heap index at level K + 2
        rr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, cb[u] + (m8[u] << 1) - (uint32_t)(32u << K), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // Rounds: every walker is at a leaf after x.max_rounds visits of this loop (the host knows the forest's deepest path), so the LAST possible
    // round needs no gather behind its visits -- until round 5 it issued one per tree all the same (nine instead of eight per tree and wave on
    // BASELINE config 4, each ~30 cycles of the vector-memory pipe: the unit this kernel is bound by).
    bool alive = true;  // some lane of the wave is still walking
    if (max_rounds > 1u) {
      uint32_t r = 1u;
      do {  // (one exit: with a counted exit beside the ballot's the structurizer copies the eight lane masks round every back edge)
        bool any = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          // the record stays opaque until its own visit: otherwise pieces of later visits are hoisted in front of the earlier gathers
          // and the first wait of a round covers half the queue
          asm volatile("" : "+v"(rr[u].x), "+v"(rr[u].y), "+v"(rr[u].z), "+v"(rr[u].w));
          const uint32_t f = sp_feature<Q, GF>(rr[u].y, lane_off, gs);
          const bool right = sp_right<SLOW, Q>(f, rr[u].x, rr[u].y, miss_key);
          const uint32_t lw = right ? (rr[u].y << 1) : rr[u].y;  // kSpRightLeaf (bit 30) or kSpLeftLeaf (bit 31) into the sign bit
          const bool leaf = (int32_t)lw < 0;
          const uint32_t nxt = right ? rr[u].w : rr[u].z;
          if (act[u] && leaf) leafv[u] = __uint_as_float(nxt);
          act[u] = act[u] && !leaf;
          any = any || act[u];
          rr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, act[u] ? (nxt << 4) : idle_off, 0, 0);
          __builtin_amdgcn_sched_barrier(0);  // or the scheduler collects the loads at the end of the round again
        }
        alive = __ballot(any) != 0ull;
      } while (alive && ++r < max_rounds);
    }
    if (!alive) {
      // the wave left early with its last round's (idle) gathers in flight: consume them here, so that both exits reach the next pass with
      // nothing outstanding in the compiler's books
#pragma unroll
      for (int u = 0; u < U; ++u) asm volatile("" : : "v"(rr[u].x), "v"(rr[u].y), "v"(rr[u].z), "v"(rr[u].w));
    }
    if (alive) {
      // the last round: visits only.  A walker that is still alive stands on a record whose taken side is a leaf; a finished one has the zeros
      // its out-of-range gather returned (no leaf flag) -- so the visit needs no `act` (a lane mask that is live out of the loop above costs
      // three scalar instructions per tree and round on its back edge).  The host grants this round only with sparse_idle_oob on.
#pragma unroll
      for (int u = 0; u < U; ++u) {
        asm volatile("" : "+v"(rr[u].x), "+v"(rr[u].y), "+v"(rr[u].z), "+v"(rr[u].w));
        const uint32_t f = sp_feature<Q, GF>(rr[u].y, lane_off, gs);
        const bool right = sp_right<SLOW, Q>(f, rr[u].x, rr[u].y, miss_key);
        const uint32_t lw = right ? (rr[u].y << 1) : rr[u].y;
        const uint32_t nxt = right ? rr[u].w : rr[u].z;
        if ((int32_t)lw < 0) leafv[u] = __uint_as_float(nxt);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

#pragma unroll
    for (int h = 0; h < U / 8; ++h) {
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
  }
}

// Stage the block's tuple tile feature-major in LDS (quad-coalesced loads + in-quad DPP transpose, see score_tile_kernel); returns the
// lane's "saw a missing value" flag.
template <int THREADS, int FEAT_OFF, int ROW>
__device__ __forceinline__ uint32_t sparse_stage_tile(const ScoreArgs& a, const uint64_t tile0, const int tid) {
  const uint32_t W = a.tuple_words, lpt = W / 4u;
  uint32_t miss_any = 0;
  const uint32_t col = (uint32_t)tid, t4 = col & 3u;
  const bool valid = tile0 + col < a.n;
  const uint64_t quad_row = tile0 + (uint64_t)(col & ~3u);
  for (uint32_t g0 = 0; g0 < lpt; g0 += 8) {
    u32x4 v[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t line = g0 + 4u * h + t4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint64_t rj = quad_row + (uint64_t)j;
        v[h][j] = (rj < a.n && line < lpt) ? *reinterpret_cast<const u32x4*>(a.tuples + rj * W + 4u * line) : u32x4{0u, 0u, 0u, 0u};
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
  return miss_any;
}

template <int K, int U, int THREADS, bool Q, bool DK, bool GF = false, int M = 0, bool PR = false>
__global__ __launch_bounds__(THREADS) void score_sparse_kernel(const ScoreArgs a, const SparseAux x) {
  static_assert(!GF || (!Q && !DK), "global-feature fallback: fp32 keys, 16-byte level K-1 records");
  constexpr int TOPB = (DK ? 8 : 12) << K;  // bytes of one tree's top image
  constexpr int STEPB = U * TOPB;  // top images resident per pass: U trees walked in lock-step = U independent load chains per lane
  constexpr int ROW = Q ? THREADS * 2 : THREADS * 4;
  constexpr int FEAT_OFF = (STEPB + ROW - 1) / ROW * ROW;
  static_assert(U == 8 || U == 16, "one or two PU groups per pass");
  static_assert((STEPB / 16) % 64 == 0, "whole waves per DMA");
  static_assert(!Q || THREADS == 1024, "the rank pre-pass writes tiles of 1024 tuples");
  const int tid = threadIdx.x;
  const uint64_t tile0 = (uint64_t)blockIdx.x * THREADS;
  const uint32_t W = a.tuple_words;

  if (!DK) dma_chunk<THREADS, STEPB>(a.img, 0, 0, tid);  // top images of the first pass (dense level K: after the missing-value test)

  bool slow;
  GfSrc gs{};
  if constexpr (GF) {
    // no tile: the walk gathers its features from the block's rows (rows past n read as 0 through the resource's range check); the
    // missing rule is applied at every visit
    const uint64_t left = a.n - tile0;
    const uint32_t rows_here = left < (uint64_t)THREADS ? (uint32_t)left : (uint32_t)THREADS;
    gs.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.tuples + tile0 * W), 0, (int)(rows_here * W * 4u), 0x00020000);
    gs.miss_raw = a.miss_raw;
    gs.ieee = a.ieee;
    slow = true;
  } else if constexpr (Q) {
    // the feature tile is one contiguous block of W * 2048 bytes of the pre-pass's output: DMA it in (score_q16_kernel); the
    // first barrier of the walk publishes it
    const uint4* src = reinterpret_cast<const uint4*>(x.q16.q + (uint64_t)blockIdx.x * W * (uint32_t)THREADS);
    const uint32_t units = W * (ROW / 16);
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    for (uint32_t u0 = 0; u0 < units; u0 += THREADS) {
      const uint32_t lds_addr = (uint32_t)FEAT_OFF + (u0 + (uint32_t)wave_base) * 16u;
      const uint4* g = src + (u0 + (uint32_t)tid);
      if (u0 + (uint32_t)wave_base < units)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds_addr), "v"(g) : "memory");
    }
    slow = __builtin_amdgcn_readfirstlane((int)x.q16.tile_flags[blockIdx.x]) != 0;  // the tile holds a missing value
  } else {
    const uint32_t miss_any = sparse_stage_tile<THREADS, FEAT_OFF, ROW>(a, tile0, tid);
    // the barrier inside publishes the staged tile; tiles without a missing value skip the missing rule altogether
    // (dense level K has no flag bytes behind the tile: its LDS is full to the byte; the flags use offset 0 before the first image)
    slow = block_any<THREADS>(miss_any, DK ? 0u : (uint32_t)FEAT_OFF + W * (uint32_t)ROW, tid);
  }
  if (DK) {
    __syncthreads();  // every wave has read the flags: the image may land on them
    dma_chunk<THREADS, STEPB>(a.img, 0, 0, tid);
  }

  RefAcc<1> ra;
  ra.init();
  double dacc = 0.0;
  const uint32_t C = a.clusters;
  if constexpr (GF) sparse_walk<K, U, THREADS, true, Q, DK, true>(a, x, tid, ra, dacc, gs);
  else if (!slow) sparse_walk<K, U, THREADS, false, Q, DK, false, M, PR>(a, x, tid, ra, dacc, gs);
  else sparse_walk<K, U, THREADS, true, Q, DK, false, M, PR>(a, x, tid, ra, dacc, gs);
  ra.align(C);
  const uint64_t row = tile0 + (uint64_t)tid;
  if (row < a.n) a.out[row] = (a.sum_mode == 1) ? (float)dacc : ra.total(0, C, a.sum_mode == 2);
}

template <int K, int U, int THREADS, bool Q, bool DK = false, bool GF = false, int M = 0, bool PR = false>
static hipError_t launch_sparse_v(const ScoreArgs& a, const Variant& v, hipStream_t s) {
  const SparseAux& x = *reinterpret_cast<const SparseAux*>(a.aux);
  const uint32_t lds = v.lds_bytes_sparse(a.tuple_words);
  auto kern = score_sparse_kernel<K, U, THREADS, Q, DK, GF, M, PR>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const uint64_t blocks = (a.n + THREADS - 1) / THREADS;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
  if (Q && !x.q16.skip_prepass) {  // ranks + per-tile missing flags of this batch (reused by the other classes' launches)
    e = launch_q16_prepass(a, x.q16, s);
    if (e != hipSuccess) return e;
  }
  if (a.ev_mid) (void)hipEventRecord(a.ev_mid, s);
  hipLaunchKernelGGL(kern, dim3((uint32_t)blocks), dim3(THREADS), lds, s, a, x);
  return hipGetLastError();
}

#define DDT_SP(K, U, T) \
  Variant { "sparse_k" #K "_u" #U "_t" #T, kKindSparse, K, T, 1, U, U, 1, 0, &launch_sparse_v<K, U, T, false> }
#define DDT_SPD(K, U, T) /* dense level K: 8-byte records only in LDS (8 * 2^K bytes per tree) */ \
  Variant { "sparse_dk_k" #K "_u" #U "_t" #T, kKindSparse, K, T, 1, U, U, 1, 2, &launch_sparse_v<K, U, T, false, true> }
#define DDT_SPM(M, K, U, T) /* dense level K+M behind M dense levels of 8-byte records (opt bit 3; Variant::top = M) */ \
  Variant { "sparse_dm" #M "_k" #K "_u" #U "_t" #T, kKindSparse, K, T, 1, U, U, 1, 2 | 8, &launch_sparse_v<K, U, T, false, true, false, M>, M }
#define DDT_SPP(K, U, T) /* dense pair records for the levels K and K+1, dense level K+2 (opt bit 4; Variant::top = 2; at most 256 features) */ \
  Variant { "sparse_dp_k" #K "_u" #U "_t" #T, kKindSparse, K, T, 1, U, U, 1, 2 | 16, &launch_sparse_v<K, U, T, false, true, false, 0, true>, 2 }
#define DDT_SPQP(K, U) /* rank-quantised + dense pair records */ \
  Variant { "sparse_qp_k" #K "_u" #U "_t1024", kKindSparse, K, 1024, 1, U, U, 1, 1 | 2 | 16, &launch_sparse_v<K, U, 1024, true, true, false, 0, true>, 2 }
#define DDT_SPQD(K, U) /* rank-quantised + dense level K */ \
  Variant { "sparse_qd_k" #K "_u" #U "_t1024", kKindSparse, K, 1024, 1, U, U, 1, 3, &launch_sparse_v<K, U, 1024, true, true> }
#define DDT_SPG(K, U, T) /* global features: no tile in LDS, any tuple width */ \
  Variant { "sparse_gf_k" #K "_u" #U "_t" #T, kKindSparse, K, T, 1, U, U, 1, 4, &launch_sparse_v<K, U, T, false, false, true> }

// `levels` = K (top levels staged in LDS), `chunk_trees` = trees walked in lock-step, `threads` = tuples per tile
static const Variant g_sparse_variants[] = {
    // rank-quantised (thresholds -> ranks, features -> the u16 tiles of the q16 pre-pass): half the LDS per tuple, so a CU holds
    // 1024 walkers = 16 waves instead of 512 = 8 -- the deep phase is latency-bound, walkers in flight are what it needs
    // (the rank-quantised kernels without the dense level K, `sparse_q_k*`, went at the end of round 6: wherever one fitted, `sparse_qd_k*` of the same K fits)
    DDT_SPQD(6, 8), DDT_SPQD(7, 8), DDT_SPQD(8, 8), DDT_SPQD(9, 8), DDT_SPQD(10, 8),
    // measured and NOT instantiated (profiles/archive/r02_sparse_sweep_*.log): 16 trees in lock-step (u16: no gain over u8), half a
    // PU group per pass (u4: K + 1 at the same occupancy, but 4 loads in flight per lane: 159 vs 196 Mtuples/s)
    // (`sparse_k*_t256` likewise: the dense-level-K kernels below take a third less LDS per tree at the same tile)
    // (512-tuple tiles -- one block of 8 waves per CU with ONE set of top images -- went in round 6: the automatic choice never took them: K = 8 in
    // two 256-tuple blocks 256.6 Mtuples/s, K = 9 in one block of 512 243.4, profiles/archive/r03_sparse_dense_level_k.json)
    // narrower tiles for wide tuples (the feature tile is 4 * W bytes per tuple)
    DDT_SP(6, 8, 128), DDT_SP(7, 8, 128), DDT_SP(8, 8, 128), DDT_SP(9, 8, 128), DDT_SP(10, 8, 128),
    DDT_SP(8, 8, 64), DDT_SP(9, 8, 64), DDT_SP(10, 8, 64),
    // dense level K: K = 8 at two 256-tuple blocks per CU / K = 9 in one block of 512 where the 16-byte level K-1 records allow 7 / 8
    DDT_SPD(6, 8, 256), DDT_SPD(7, 8, 256), DDT_SPD(8, 8, 256), DDT_SPD(9, 8, 256), DDT_SPD(10, 8, 256),
    // dense mid levels (round 5): forests whose levels right below the top image are (nearly) complete
    // (BASELINE config 4, one box, alternating: M = 0 / 1 / 2 / 3 -> 265-270 / 275 / 271-272 / 214-227 Mtuples/s, profiles/r05_pmc_cfg2_cfg4_cfg6.md:
    // one mid level pays a little, three lose a fifth -- the padding under the early leaves of level 10 turns finished walkers into live gathers)
    // (the 128-tuple forms went in round 6: no dense-level-K kernel of that tile exists for them to be the sibling of)
    DDT_SPM(1, 8, 8, 256), DDT_SPM(1, 7, 8, 256),  // (two mid levels, `sparse_dm2_k8`: +1 % where one gives +2-4 %, never the automatic choice: removed)
    // dense pair records (round 5): two levels per gather below the top image, for forests that fill the levels K .. K+2
    // (K = 10 measured and NOT instantiated: the dense block would sit at level 12, 64 KiB per tree -- a 255-bin version of config 4 on 32 features:
    // `sparse_qp_k10` 272.8 vs `sparse_qd_k10` 287.1 Mtuples/s; K = 9 on 64 features: 267.4 vs 263.1)
    // (one block of 512 tuples per CU, measured and NOT instantiated: `sparse_dp_k9_u8_t512` 250.5, `sparse_dp_k8_u8_t512` 258.0 against 293.4 Mtuples/s)
    DDT_SPP(7, 8, 256), DDT_SPP(8, 8, 256), DDT_SPP(9, 8, 256),
    DDT_SPQP(7, 8), DDT_SPQP(8, 8), DDT_SPQP(9, 8),
    // tuples too wide for any feature tile (more than ~540 words): every feature is gathered from the tuple's row in global memory.
    // The correctness path of the sparse format, like the generic kernel of the perfect-tree format -- not a tuned kernel.
    DDT_SPG(6, 8, 256),
};

int num_sparse_variants() { return (int)(sizeof(g_sparse_variants) / sizeof(g_sparse_variants[0])); }
const Variant& sparse_variant(int i) { return g_sparse_variants[i]; }

}  // namespace ddt
