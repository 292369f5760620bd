// ddt_deep.hip -- scoring kernel for DEEP perfect trees (depth 9..15): the reference's own example configuration is 512 trees of depth 12
// over 32 features (profiler/profiler.cpp:32-38; a depth-12 tree is exactly one PU's 8192 words, rtl/DTEngine/core/DTPU.sv:22-25).
//
// What it replaces: the same DTPU traversal loop + leaf reduce as ddt_kernels.hip (DTPU.sv:579-760; FPAddersReduceTree.sv:94-141,
// FPAggregator.v:79-131, Core.sv:486-541) for trees whose node memory no longer fits a CU's LDS next to the tuples at two blocks per CU.
// Rank-quantised like the depth-8 kernels (same records, same u16 rank tile, same pre-pass: ddt_prepass.hip launch_q16_prepass); image
// layout: ddt_internal.h "deep rank-quantised kernels"; host packing: ddt_image.cpp pack_image_q16.  No MFMA: compare + gather.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ddt_device.h"
#include "ddt_internal.h"

namespace ddt {

// ---------------------------------------------------------------------------------------------------
// "q16d": perfect trees DEEPER than a CU's LDS holds at two blocks per CU -- the reference's own example configuration is 512 trees of
// depth 12 (profiler/profiler.cpp:32-38; a depth-12 tree is exactly one PU's 8192 words, DTPU.sv:22-25).  The first K = 8 / 9 levels of a
// tree are walked out of LDS exactly like the depth-8 kernels (4-byte records, u16 rank tile, 2 x 1024 threads per CU); the D - K levels
// below come from the image in global memory, (D - K + 1) / 2 GATHERS of one 16-byte record per tree and tuple (ddt_internal.h): a "pair"
// record {node, left child, right child, offset of the next block} decides two levels, the "terminal" record {node, left leaf, right leaf}
// the last level and the leaf -- depth 12 with K = 9 costs two gathers per tree, where a record-per-level walk costs four (one per
// level + the leaf).  The vector-memory pipe takes about one lane address per cycle and CU whatever the bytes (the
// sparse kernel's bound), so gathers per tree are what to minimise.
// The gathers of a sub-group of four trees are a ROTATING PIPELINE across sub-groups: sub-group s issues its first gather right after its
// LDS walk; one sub-group later (the LDS walk of s + 1 in between) the record is used and the next gather goes out; the leaves are
// folded G sub-groups late, in sub-group order, so the reference's order of adds is kept (FPAddersReduceTree.sv:94-141,
// FPAggregator.v:79-131, Core.sv:486-541 through the cluster-major running total of the "_cm" kernels).  The pipeline runs across chunk
// barriers: the gathers do not touch the chunk buffers, and the barrier's wait counts them out (vmcnt(N): the DMA of the next chunk is older
// than the N gathers issued behind it; loads return in order).
// ---------------------------------------------------------------------------------------------------
template <int FEAT_OFF, bool SLOW, bool WIDE = false>
__device__ __forceinline__ uint32_t q16_rank_at(uint32_t rec, uint32_t lane2) {
  uint32_t off = SLOW ? ((rec >> 16) & 0xFFFEu) : (rec >> 16);
  if (WIDE) off <<= 1;
  return *reinterpret_cast<const DDT_LDS(uint16_t)*>((off | lane2) + (uint32_t)FEAT_OFF);
}
template <bool SLOW>
__device__ __forceinline__ bool q16_goes_right(uint32_t rec, uint32_t fv) {
  bool right = fv >= (rec & 0xFFFFu);
  if (SLOW) right = (fv == kQMissing) ? ((rec >> 16) & 1u) != 0u : right;
  return right;
}
constexpr uint32_t q16d_stage_level(int D, int K, int g) { return g + 1 < (D - K + 1) / 2 ? (uint32_t)(K + 2 * g) : (uint32_t)(D - 1); }
constexpr uint32_t q16d_stage_off(int D, int K, int g) {
  uint32_t off = 0;
  for (int k = 0; k < g; ++k) off += 16u << q16d_stage_level(D, K, k);
  return off;
}

// (three or four gathers per tree in flight -- depths 13..15 -- need more than the 64 VGPRs of two blocks per CU: one block of 16 waves then)
// SPLIT (small batches, as score_q16_kernel's): grid (tiles, slices); block (tile, s) walks split_len PU groups of this launch's image from a zero
// accumulator and leaves every group's sum (its 8-leaf reduce tree + 0) in out[group0 + the group's place in the image][row]; the adds follow in
// launch_cm_combine, in the reference's order -- so the parts of an ensemble scored in parts need no state handed from launch to launch either.
// (One block of 16 waves per CU: the stores keep the row index alive, and a batch of a few tiles has no second block to place.)
template <int D, int K, int CT, bool WIDE = false, bool SPLIT = false>
__global__ __launch_bounds__(kQTile, ((D - K + 1) / 2 >= 3 || WIDE || SPLIT) ? 4 : 8) void score_q16d_kernel(const ScoreArgs a, const Q16Aux x) {
  constexpr int THREADS = kQTile, U = 4;
  constexpr int G = (D - K + 1) / 2;  // gathers per tree
  static_assert((D - K) % 2 == 1 && G >= 1 && G <= 4 && K >= 3, "D - K odd: pair stages and one terminal stage");
  static_assert(CT % U == 0 && (CT == 4 || CT == 8), "a chunk = half a PU group or a whole one");
  constexpr int TOPB = 4 << K;                      // a tree's records in LDS
  constexpr int CHUNK_BYTES = TOPB * CT;
  constexpr uint32_t DEEPB = q16d_stage_off(D, K, G);  // a tree's stage blocks in the global image
  constexpr uint32_t GCHUNK = (uint32_t)CT * ((uint32_t)TOPB + DEEPB);
  constexpr int GSKIP = (int)((GCHUNK - (uint32_t)CHUNK_BYTES) / 16u);
  constexpr int FEAT_OFF = 2 * CHUNK_BYTES;
  constexpr int ROW = kQTile * 2;
  constexpr int SGS = CT / U;                       // sub-groups per chunk
  static_assert(FEAT_OFF % ROW == 0, "row|lane OR trick");
  const int tid = threadIdx.x;
  const uint64_t tile = blockIdx.x, tile0 = tile * kQTile;
  const uint32_t W = a.tuple_words;
  uint32_t n_chunks = a.n_chunks;  // whole PU groups (an ensemble's part ends on a whole group too)
  const bool slow = __builtin_amdgcn_readfirstlane((int)x.tile_flags[tile]) != 0;
  const uint4* img = slow ? x.img_slow : a.img;
  uint32_t sp_pos = 0;
  const uint64_t sp_row = tile0 + (uint64_t)tid;
  if constexpr (SPLIT) {
    constexpr uint32_t CPG = 8u / (uint32_t)CT;  // chunks per PU group (CT = 4: two, CT = 8: one)
    const uint32_t groups_here = a.n_chunks / CPG, first = blockIdx.y * x.split_len;
    n_chunks = (groups_here - first < x.split_len ? groups_here - first : x.split_len) * CPG;  // >= 1 group: the host launches ceil(real groups / split_len) slices
    img += (size_t)first * CPG * (size_t)(GCHUNK / 16u);
    sp_pos = x.group0 + first;
  }

  dma_chunk<THREADS, CHUNK_BYTES>(img, 0, 0, tid);
  {  // the rank tile: one contiguous block of W * 2048 bytes (score_q16_kernel)
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
  // cluster-major accumulate (score_q16_kernel "_cm"): one accumulator + a running total; a later part of an ensemble scored in parts takes
  // up the sum where the launch before left it
  RefAcc<1> ra;
  ra.init();
  double dacc[1] = {0.0};
  const uint32_t Cc = a.clusters, lane2 = (((uint32_t)tid & 511u) << 2) | (((uint32_t)tid >> 9) << 1);
  const uint32_t cm_lg = (uint32_t)__builtin_ctz(Cc | 0x100u), cm_real = x.real_groups;
  uint32_t cm_groups = 0, cm_cluster = 0, cm_bound = SPLIT ? 1u : (cm_real + Cc - 1u) >> cm_lg;
  float cm_total = 0.f;
  if constexpr (!SPLIT) {
    if (x.group0) {
      cm_groups = x.group0;
      while (cm_cluster < Cc && cm_groups >= cm_bound) {
        ++cm_cluster;
        cm_bound += cm_cluster < Cc ? (cm_real + Cc - 1u - cm_cluster) >> cm_lg : 0u;
      }
    }
    if (x.state_in) {
      const uint64_t r0 = tile0 + (uint64_t)tid;
      ra.a[0][0] = x.state_in[r0];
      cm_total = x.state_in[x.n_pad + r0];
    }
  }
  const bool exact = a.sum_mode == 2u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(img), 0, (int)(n_chunks * GCHUNK), 0x00020000);

  // the pipeline: pend[g][u] = the stage-g record of tree u of the sub-group g + 1 behind the one being walked; sbase[g] = byte offset of
  // that sub-group's stage blocks in the image (wave-uniform)
  u32x4 pend[G][U];
  uint32_t sbase[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    sbase[g] = 0u;
#pragma unroll
    for (int u = 0; u < U; ++u) pend[g][u] = u32x4{0u, 0u, 0u, 0u};
  }

  // one sub-group step.  TOPWALK: walk sub-group (k, SG) out of LDS buffer BUF and issue its first gather; then every sub-group in flight
  // advances one stage (oldest first: their records arrived longest ago) and the oldest one's leaves are folded.  DRAIN = j: the j-th step
  // behind the last sub-group -- only the sub-groups that still exist advance.
  auto step = [&](auto slow_tag, auto buf_tag, auto sg_tag, auto walk_tag, auto drain_tag, const uint32_t k, const uint32_t s_index) {
    constexpr bool SLOW = decltype(slow_tag)::value, TOPWALK = decltype(walk_tag)::value;
    constexpr int BUF = decltype(buf_tag)::value, SG = decltype(sg_tag)::value, DRAIN = decltype(drain_tag)::value;
    // ---- this sub-group: K levels out of LDS (the gathers issued by the step before fly meanwhile) ----
    uint32_t m4[U];
    if constexpr (TOPWALK) {
#pragma unroll
      for (int u = 0; u < U; ++u) m4[u] = 4u;
#pragma unroll
      for (int lvl = 0; lvl < K; ++lvl) {
        uint32_t nd[U], f[U];
#pragma unroll
        for (int u = 0; u < U; ++u) nd[u] = lds_u32(m4[u] + (uint32_t)(BUF * CHUNK_BYTES + (SG * U + u) * TOPB));
#pragma unroll
        for (int u = 0; u < U; ++u) f[u] = q16_rank_at<FEAT_OFF, SLOW, WIDE>(nd[u], lane2);
#pragma unroll
        for (int u = 0; u < U; ++u) m4[u] = (m4[u] << 1) + (q16_goes_right<SLOW>(nd[u], f[u]) ? 4u : 0u);
      }
    }
    // nothing below may be scheduled in front of the walk: the first use of a gathered record is where hipcc puts its vmcnt wait, and
    // the walk is the time the gathers have to arrive
    __builtin_amdgcn_sched_barrier(0);
    // ---- the oldest sub-group (s - G): terminal record -> leaves -> fold ----
    {
      float lf[1][U];
      uint32_t f[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // (the unused fourth word is kept alive up to here: hipcc otherwise narrows the load to three registers and hands the fourth to the
        // walk as a temporary -- a write to the destination of a load in flight, i.e. a vmcnt wait at the top of every walk)
        asm volatile("" : : "v"(pend[G - 1][u].w));
        f[u] = q16_rank_at<FEAT_OFF, SLOW, WIDE>(pend[G - 1][u].x, lane2);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) lf[0][u] = __uint_as_float(q16_goes_right<SLOW>(pend[G - 1][u].x, f[u]) ? pend[G - 1][u].z : pend[G - 1][u].y);
      if (s_index >= (uint32_t)G) {  // wave-uniform: the first G steps have nothing to fold
        constexpr int PH = ((TOPWALK ? SG + (SGS == 1 ? BUF : 0) : DRAIN) + G) & 1;  // parity of sub-group s - G (s is even at every chunk pair and in front of the drain)
        fold_leaves<U, 1, 0>(lf, PH, 1u, ra, dacc, exact);
        if (PH == 1) {  // a PU group is complete
          if constexpr (SPLIT) {  // the group's sum goes out as it is
            a.out[(uint64_t)sp_pos * x.n_pad + sp_row] = ra.a[0][0];
            ra.a[0][0] = 0.f;
            ++sp_pos;
          } else if (++cm_groups == cm_bound) {  // ... and it was its cluster's last
            cm_total = exact ? radd_exact(ra.a[0][0], cm_total) : ra.a[0][0] + cm_total;
            ra.a[0][0] = 0.f;
            ++cm_cluster;
            cm_bound += cm_cluster < Cc ? (cm_real + Cc - 1u - cm_cluster) >> cm_lg : 0u;
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- sub-groups s - G + 1 .. s - 1 (oldest first): pair record -> two levels -> the next stage's gather.  The four trees advance
    //      together (four rank reads in flight per level), then the four gathers go out back to back ----
#pragma unroll
    for (int g = G - 1; g >= 1; --g) {
      if (DRAIN >= g) continue;  // (drain: that sub-group does not exist)
      uint32_t f0[U], f1[U], c[U], voff[U];
#pragma unroll
      for (int u = 0; u < U; ++u) f0[u] = q16_rank_at<FEAT_OFF, SLOW, WIDE>(pend[g - 1][u].x, lane2);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool r0 = q16_goes_right<SLOW>(pend[g - 1][u].x, f0[u]);
        c[u] = r0 ? pend[g - 1][u].z : pend[g - 1][u].y;
        voff[u] = pend[g - 1][u].w + (r0 ? 32u : 0u);
        f1[u] = q16_rank_at<FEAT_OFF, SLOW, WIDE>(c[u], lane2);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) voff[u] += q16_goes_right<SLOW>(c[u], f1[u]) ? 16u : 0u;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        pend[g][u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[u], sbase[g - 1] + (uint32_t)u * DEEPB + q16d_stage_off(D, K, g), 0);
        __builtin_amdgcn_sched_barrier(0);  // keep the gathers where they are issued (ddt_sparse.hip)
      }
      sbase[g] = sbase[g - 1];
    }
    // ---- this sub-group's first gather: m4 = 4 * heap index h of its level-K node, h in [2^K, 2^(K+1)) -> record h - 2^K of stage 0 ----
    if constexpr (TOPWALK) {
      const uint32_t base = k * GCHUNK + (uint32_t)(CT * TOPB) + (uint32_t)(SG * U) * DEEPB;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        pend[0][u] = __builtin_amdgcn_raw_buffer_load_b128(rs, m4[u] << 2, base + (uint32_t)u * DEEPB - (16u << K), 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      sbase[0] = base;
    }
  };

  // s_waitcnt vmcnt(N), N = the gathers a chunk's steps issue behind its DMA (the counter's field is 6 bits: N <= 63)
  auto wait_for_dma = [&]() {
    constexpr int N = SGS * G * U;
    static_assert(N == 4 || N == 8 || N == 12 || N == 16 || N == 24 || N == 32, "vmcnt immediate");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  };
  auto run = [&](auto slow_tag) {
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    uint32_t s_index = 0;
    // the first chunk and the rank tile have nothing behind them: a full wait, OUTSIDE the loop -- tools/check_dma_waits.py follows every path of
    // the built binary to the counted waits below, and the path "kernel entry -> counted wait" exists in the control-flow graph whatever k is
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t k = 0; k < n_chunks; k += 2) {
      // the DMA of this chunk is older than the SGS * G * U gathers issued behind it since the last barrier (every step issues all of its
      // gathers, valid or not): count them out
      if (k != 0u) wait_for_dma();
      __syncthreads();
      const bool more1 = k + 1 < n_chunks;  // (chunks of 4 trees come in pairs: whole PU groups; chunks of 8 may end here)
      if (more1) dma_chunk<THREADS, CHUNK_BYTES>(img + (size_t)(k + 1) * GSKIP, k + 1, CHUNK_BYTES, tid);
      step(slow_tag, I0{}, I0{}, std::true_type{}, I0{}, k, s_index++);
      if constexpr (SGS == 2) step(slow_tag, I0{}, I1{}, std::true_type{}, I0{}, k, s_index++);
      if (!more1) break;
      wait_for_dma();
      __syncthreads();
      if (k + 2 < n_chunks) dma_chunk<THREADS, CHUNK_BYTES>(img + (size_t)(k + 2) * GSKIP, k + 2, 0, tid);
      step(slow_tag, I1{}, I0{}, std::true_type{}, I0{}, k + 1, s_index++);
      if constexpr (SGS == 2) step(slow_tag, I1{}, I1{}, std::true_type{}, I0{}, k + 1, s_index++);
    }
    // drain: the sub-groups still in flight
    step(slow_tag, I0{}, I0{}, std::false_type{}, I0{}, 0u, s_index++);
    if constexpr (G >= 2) step(slow_tag, I0{}, I0{}, std::false_type{}, I1{}, 0u, s_index++);
    if constexpr (G >= 3) step(slow_tag, I0{}, I0{}, std::false_type{}, std::integral_constant<int, 2>{}, 0u, s_index++);
    if constexpr (G >= 4) step(slow_tag, I0{}, I0{}, std::false_type{}, std::integral_constant<int, 3>{}, 0u, s_index++);
  };
  if (!slow) run(std::false_type{});
  else run(std::true_type{});

  if constexpr (SPLIT) return;  // (every group's sum went out at its fold)
  // (the row index is recomputed here from an opaque copy of the thread id: kept across the walk it costs the 64th and 65th VGPR, i.e. a spill)
  uint32_t tid_end = (uint32_t)threadIdx.x;
  asm volatile("" : "+v"(tid_end));
  const uint64_t row = (uint64_t)blockIdx.x * kQTile + (uint64_t)tid_end;
  if (x.state_out) {  // not the ensemble's last part: the sum's state instead of the score
    x.state_out[row] = ra.a[0][0];
    x.state_out[x.n_pad + row] = cm_total;
    return;
  }
  if (row < a.n) a.out[row] = cm_total;
}

template <int D, int K, int CT, bool WIDE = false>
static hipError_t launch_q16d(const ScoreArgs& a, const Variant& v, hipStream_t s) {
  const Q16Aux& x = *reinterpret_cast<const Q16Aux*>(a.aux);
  const uint64_t tiles = (a.n + kQTile - 1) / kQTile;
  if (tiles == 0) return hipSuccess;
  if (tiles > 0x7FFFFFFFull || (CT == 4 && (a.n_chunks & 1u)) || a.sum_mode == 1u) return hipErrorInvalidValue;
  auto kern = score_q16d_kernel<D, K, CT, WIDE>;
  const uint32_t lds = v.lds_bytes_q16(a.tuple_words);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if (!x.skip_prepass) {
    e = launch_q16_prepass(a, x, s);
    if (e != hipSuccess) return e;
  }
  if (a.ev_mid) (void)hipEventRecord(a.ev_mid, s);
  if (x.split > 0u) {  // a small batch cut into slices of PU groups (launch_score decides; every deep kernel has the form)
    auto ksplit = score_q16d_kernel<D, K, CT, WIDE, true>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ksplit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ksplit, dim3((uint32_t)tiles, x.split), dim3(kQTile), lds, s, a, x);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(kern, dim3((uint32_t)tiles), dim3(kQTile), lds, s, a, x);
  return hipGetLastError();
}


#define DDT_QD(NAME, D, K, CT) \
  Variant { NAME, kKindQ16, D, kQTile, 1, CT, 4, 1, 36, &launch_q16d<D, K, CT>, K }
#define DDT_QDW(NAME, D, K, CT) /* wide tuples (33..64 words): one block per CU */ \
  Variant { NAME, kKindQ16, D, kQTile, 1, CT, 4, 1, 36 | 64, &launch_q16d<D, K, CT, true>, K }

// opt = cluster-major | deep (| wide), last field = K.  Table order = the engine's order of preference (ddt_choice.cpp auto_variant): the
// two-blocks-per-CU forms first, then the wide ones for tuples of 33..64 words
static const Variant g_deep_variants[] = {
    DDT_QD("q16d_d12_k9_c4_u4_cm", 12, 9, 4),
    DDT_QD("q16d_d11_k8_c8_u4_cm", 11, 8, 8),
    DDT_QD("q16d_d10_k9_c4_u4_cm", 10, 9, 4),
    DDT_QD("q16d_d9_k8_c8_u4_cm", 9, 8, 8),
    DDT_QD("q16d_d13_k8_c8_u4_cm", 13, 8, 8),
    DDT_QD("q16d_d14_k9_c4_u4_cm", 14, 9, 4),
    DDT_QD("q16d_d15_k8_c8_u4_cm", 15, 8, 8),
    // (eleven levels out of LDS and ONE gather per tree, `q16d_d12_k11_c4`: 2 x 32 KiB of records + the rank tile = one block per CU -- measured and
    // NOT instantiated: 512 / 171 / 64 trees x d12, 10 M tuples: 14.97 / 5.52 / 2.29 ms against 14.31 / 5.22 / 2.24 of the two-gather, two-block form)
    // (depth 16 measured and NOT instantiated: 64 trees x 32 features, 4 M tuples -- 693 Mtuples/s against 725 on the generic kernel: 131 k
    // thresholds per feature = four parts, each with a transpose + rank pre-pass, and four gathers per tree at one block per CU)
    DDT_QDW("q16dw_d12_k9_c4_u4_cm", 12, 9, 4),
    DDT_QDW("q16dw_d11_k8_c8_u4_cm", 11, 8, 8),
    DDT_QDW("q16dw_d10_k9_c4_u4_cm", 10, 9, 4),
    DDT_QDW("q16dw_d9_k8_c8_u4_cm", 9, 8, 8),
};

int num_deep_variants() { return (int)(sizeof(g_deep_variants) / sizeof(g_deep_variants[0])); }
const Variant& deep_variant(int i) { return g_deep_variants[i]; }

}  // namespace ddt
