// ddt_prepass.hip -- the rank pre-pass of the rank-quantised ("q16") path, split out of ddt_kernels.hip in round 6: the transposes
// (plain, and the gathering one of feature compaction), rank_kernel (tables of up to 32767 keys per feature, one feature per block) and the
// LDS-resident forms that need no transposed intermediate (fused_rank_kernel: all tables resident together; grouped_rank_kernel: feature
// groups x row partitions).  What they replace in the reference: nothing -- its PUs compare raw fp32 words (DTPU.sv:653-657); the ranks
// are an exact re-encoding of that compare (r(x) >= R  <=>  !(x < t)), held to it by tests/test_rank_transform.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ddt_device.h"
#include "ddt_internal.h"

namespace ddt {

// ---------------------------------------------------------------------------------------------------
// The rank-quantised path ("q16").  A node test only needs the ORDER of x[fidx] relative to the node's
// threshold, so a feature can be replaced EXACTLY by its rank among the sorted distinct thresholds the model
// uses on that feature:  r(x) = #{t_k <= x},  node rank R = k + 1  =>  !(x < t_k)  <=>  r(x) >= R.
// Ranks fit 16 bits (the engine falls back to the fp32 kernels otherwise), which halves the feature tile
// (64 KiB per 1024 tuples) and the node records (4 bytes {R, row offset}): two 1024-thread blocks fit a CU,
// 32 waves instead of 16 -- the occupancy the fp32 tile cannot reach (measured +30 % node-visits/s).
// Three launches per batch:
//   transpose_kernel  tuples [n][W] fp32  ->  xT [W][n_pad]                           (HBM streaming)
//   rank_kernel       per feature: threshold table in LDS, branch-free binary search  ->  q [tile][W][1024] u16,
//                     missing -> 0xFFFF and tile_flags[tile] = 1
//   score_q16_kernel  the walk over u16 ranks; the tile arrives by global->LDS DMA, no transpose needed
// ---------------------------------------------------------------------------------------------------
// (kQTile = 1024, tuples per q tile == threads per scoring block: ddt_internal.h)
constexpr uint32_t kRankBuckets = kQ16RankBuckets;  // slices of a feature's key range (first level of the rank search)

__global__ __launch_bounds__(256) void transpose_kernel(const uint32_t* __restrict__ tuples, uint32_t W, uint64_t n,
                                                        uint64_t n_pad, uint32_t* __restrict__ xT) {
  // 256 rows per block through LDS [256][W+1] (odd stride: conflict-free column reads)
  const uint32_t tid = threadIdx.x, S = W + 1u;
  const uint64_t row0 = (uint64_t)blockIdx.x * 256u;
  const uint32_t rows = row0 >= n ? 0u : (uint32_t)((n - row0) < 256u ? (n - row0) : 256u);  // blocks past n only write the zero padding
  const uint4* src = reinterpret_cast<const uint4*>(tuples + row0 * W);
  const uint32_t LPT = W / 4u;  // 16-byte lines per tuple; eight at a time (the perfect-tree q16 path has <= 8, sparse forests up to 19)
  for (uint32_t i0 = 0; i0 < LPT; i0 += 8u) {
    uint4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // all loads first (coalesced, independent), then the LDS scatter
      const uint32_t k = i0 + (uint32_t)i, e = tid + k * 256u;
      v[i] = (k < LPT && e / LPT < rows) ? src[e] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t k = i0 + (uint32_t)i;
      if (k < LPT) {
        const uint32_t e = tid + k * 256u, r = e / LPT, c = (e - r * LPT) * 4u;
        lds_st_u32((r * S + c + 0u) * 4u, v[i].x);
        lds_st_u32((r * S + c + 1u) * 4u, v[i].y);
        lds_st_u32((r * S + c + 2u) * 4u, v[i].z);
        lds_st_u32((r * S + c + 3u) * 4u, v[i].w);
      }
    }
  }
  __syncthreads();
  for (uint32_t c = 0; c < W; ++c) xT[(uint64_t)c * n_pad + row0 + tid] = lds_u32((tid * S + c) * 4u);
}

// Feature compaction (ddt_choice.cpp plan_feature_compaction): the same transposed intermediate from rows of `Win` words of which only the columns
// cols[0 .. Wc) are wanted (~0 = a padding column: zeros).  R rows per block go through LDS [R][Win + 1] (coalesced 16-byte loads of whole rows --
// the rows are read once, like every tuple row of every path -- odd stride: conflict-free column reads), then thread (row r, column group) writes
// its columns.  R = the largest power of two <= 64 whose stage fits 96 KiB (Win = 2048: 8 rows).
__global__ __launch_bounds__(256) void gather_transpose_kernel(const uint32_t* __restrict__ tuples, uint32_t Win, const uint32_t* __restrict__ cols,
                                                               uint32_t Wc, uint64_t n, uint64_t n_pad, uint32_t R, uint32_t* __restrict__ xT) {
  const uint32_t tid = threadIdx.x, S = Win + 1u, LPT = Win / 4u;
  const uint64_t row0 = (uint64_t)blockIdx.x * R;
  const uint32_t rows = row0 >= n ? 0u : (uint32_t)((n - row0) < R ? (n - row0) : R);
  const uint4* src = reinterpret_cast<const uint4*>(tuples + row0 * Win);
  for (uint32_t e = tid; e < rows * LPT; e += 256u) {
    const uint4 v = src[e];
    const uint32_t r = e / LPT, c = (e - r * LPT) * 4u;
    lds_st_u32((r * S + c + 0u) * 4u, v.x);
    lds_st_u32((r * S + c + 1u) * 4u, v.y);
    lds_st_u32((r * S + c + 2u) * 4u, v.z);
    lds_st_u32((r * S + c + 3u) * 4u, v.w);
  }
  __syncthreads();
  const uint32_t r = tid % R, g = tid / R, G = 256u / R;
  for (uint32_t c = g; c < Wc; c += G) {
    const uint32_t f = cols[c];
    xT[(uint64_t)c * n_pad + row0 + r] = (r < rows && f != 0xFFFFFFFFu) ? lds_u32((r * S + f) * 4u) : 0u;
  }
}

hipError_t launch_gather_transpose(const uint32_t* tuples, uint32_t Win, const uint32_t* cols, uint32_t Wc, uint64_t n, uint64_t n_pad, uint32_t* xT,
                                   hipStream_t s) {
  uint32_t R = 64;
  while (R > 8u && (size_t)R * (Win + 1u) * 4u > 96u * 1024u) R >>= 1;
  const uint32_t lds = R * (Win + 1u) * 4u;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gather_transpose_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gather_transpose_kernel, dim3((uint32_t)(n_pad / R)), dim3(256), lds, s, tuples, Win, cols, Wc, n, n_pad, R, xT);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ p, uint64_t words) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < words; i += (uint64_t)gridDim.x * 256u) p[i] = 0u;
}
hipError_t launch_zero_words(uint32_t* p, uint64_t words, hipStream_t s) {
  if (words == 0) return hipSuccess;
  const uint64_t blocks = (words + 255u) / 256u;
  hipLaunchKernelGGL(zero_words_kernel, dim3((uint32_t)(blocks < 1024u ? blocks : 1024u)), dim3(256), 0, s, p, words);
  return hipGetLastError();
}

hipError_t launch_transpose(const uint32_t* tuples, uint32_t W, uint64_t n, uint64_t n_pad, uint32_t* xT, hipStream_t s) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(transpose_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((W + 1) * 256 * 4));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(transpose_kernel, dim3((uint32_t)(n_pad / 256)), dim3(256), (W + 1) * 256 * 4, s, tuples, W, n, n_pad, xT);
  return hipGetLastError();
}

constexpr uint32_t kRankThreads = 1024;  // two blocks per CU share the LDS with their tables: 32 waves x 4 searches each
// Search = one bucket lookup + a short binary search.  The key range [lo, hi] of the feature's table is cut into
// kRankBuckets equal slices of 2^shift codes; starts[b] = number of keys in slices < b, and no slice holds P or more
// keys (P = power of two, per feature, from the host: ddt_image.cpp build_image_q16), so log2(P) probes from
// starts[b] finish the count -- keys past the slice are > x by construction, no end test.  1000 trees x 255
// nodes over 32 features (~8 k keys per table): 1 + 5 LDS reads instead of 13.  Degenerate key distributions
// only make P larger, up to the plain binary search over the whole table.
// Round 6 (late): the search as in rank_line below -- this kernel had kept round 2's form: a table skewed by i / 32 (against the bank pattern of a
// binary search from entry 0, which the bucket starts ended), every probe's index clamped, skewed and scaled = 8 VALU instructions per probe, ~75
// per value: 0.85 ms per 320 M values where the HBM needs 0.35.  Now the position is the BYTE ADDRESS of its entry in a LINEAR table, a probe is
// ds_read_b32 with (step - 1) * 4 as the DS immediate and costs compare + select + add; the clamp (one v_min against a wave-uniform bound) only runs
// for a feature whose pads behind the keys are fewer than P (K + P > Kpad, wave-uniform: the host pads long tables by 64 entries for that).  The LDS the
// skew took (Kpad / 32 words) is table now: q16_rank_lds_bytes / kQ16MaxTable (ddt_engine_priv.h).
template <bool IEEE>
__global__ __launch_bounds__(kRankThreads) void rank_kernel(const uint32_t* __restrict__ xT, uint64_t n, uint64_t n_pad,
                                                   const uint32_t* __restrict__ tables, uint32_t Kpad,
                                                   const uint32_t* __restrict__ tabP, const uint16_t* __restrict__ tabS,
                                                   uint32_t miss_raw,
                                                   uint32_t W, uint16_t* __restrict__ q, uint32_t* __restrict__ tile_flags) {
  const uint32_t j = blockIdx.y, tid = threadIdx.x;
  for (uint32_t i = tid; i < Kpad; i += kRankThreads) lds_st_u32(i * 4u, tables[(size_t)j * Kpad + i]);
  const uint32_t starts_off = Kpad * 4u;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(tabS + (size_t)j * kRankBuckets);
    for (uint32_t i = tid; i < kRankBuckets / 2u; i += kRankThreads) lds_st_u32(starts_off + i * 4u, src[i]);
  }
  __syncthreads();
  const uint32_t K = tabP[j * 8u + 0u], lo = tabP[j * 8u + 1u], hi = tabP[j * 8u + 2u], shift = tabP[j * 8u + 3u];
  const uint32_t P = tabP[j * 8u + 4u];
  const uint32_t last = (Kpad - 1u) * 4u;  // byte address of the table's last entry: always an INT_MAX pad
  // a search that starts at entry s <= K probes entries <= s + P - 2: inside the pads when K + P <= Kpad
  const bool clamp = __builtin_amdgcn_readfirstlane((int)(K + P > Kpad)) != 0;
  // Lane (tid & 511) of half-block (tid >> 9) owns tuples t and t+512 of a tile -- the two halves of one dword of
  // the q tile (see the layout note below) -- in two tiles per pass: 4 independent searches per lane (the
  // dependent LDS reads of one search are latency bound) and full 4-byte, fully coalesced stores of the ranks.
  constexpr int ILP = 4;
  const uint32_t t = tid & 511u, sub = tid >> 9;
  const uint64_t tiles = n_pad / kQTile;
  uint32_t* __restrict__ q32 = reinterpret_cast<uint32_t*>(q);
  // the column values of the NEXT pass are loaded while this pass searches (a pass is otherwise a serial chain:
  // HBM read -> 6 dependent LDS reads -> store)
  const uint64_t pass_stride = (uint64_t)gridDim.x * 4u;
  auto load_pass = [&](uint32_t (&dst)[ILP], uint64_t tile0) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {  // i = 2 * (tile within the pass) + half
      const uint64_t tile = tile0 + 2u * (uint32_t)(i >> 1);
      dst[i] = tile < tiles ? xT[(uint64_t)j * n_pad + tile * kQTile + t + 512u * (uint32_t)(i & 1)] : 0u;
    }
  };
  auto probes = [&](auto clamp_tag, const int32_t (&x)[ILP], uint32_t (&pos)[ILP]) {
    constexpr bool CL = decltype(clamp_tag)::value;
    for (uint32_t step = P >> 1; step >= 64u; step >>= 1) {  // (buckets of 128 keys and more: degenerate key distributions only)
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
        const uint32_t bound = last - (step - 1u) * 4u;  // (step < P <= 2 K and K < Kpad: no wrap)
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
          const uint32_t a = CL ? (pos[i] < bound ? pos[i] : bound) : pos[i];
          if ((int32_t)lds_u32(a + (step - 1u) * 4u) <= x[i]) pos[i] += step * 4u;
        }
      }
    }
  };
  uint32_t raw_next[ILP];
  load_pass(raw_next, (uint64_t)blockIdx.x * 4u + sub);
  for (uint64_t tile0 = (uint64_t)blockIdx.x * 4u + sub; tile0 < tiles; tile0 += pass_stride) {
    uint32_t raw[ILP], pos[ILP];
    int32_t x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) raw[i] = raw_next[i];
    load_pass(raw_next, tile0 + pass_stride);
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      x[i] = (int32_t)(IEEE ? ieee_key(raw[i]) : raw[i]);
      uint32_t b = ((uint32_t)x[i] - lo) >> shift;  // wraps to a huge value below lo: selected away next
      b = b < kRankBuckets - 1u ? b : kRankBuckets - 1u;
      b = x[i] < (int32_t)lo ? 0u : b;
      pos[i] = 4u * (uint32_t)*reinterpret_cast<const DDT_LDS(uint16_t)*>(starts_off + b * 2u);  // byte address of the bucket's first key
    }
    if (clamp) probes(std::true_type{}, x, pos);
    else probes(std::false_type{}, x, pos);
    uint32_t r[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      const uint64_t tile = tile0 + 2u * (uint32_t)(i >> 1);
      const uint64_t row = tile * kQTile + t + 512u * (uint32_t)(i & 1);
      const uint32_t p = pos[i] >> 2;
      r[i] = x[i] > (int32_t)hi ? K : (p < K ? p : K);
      if (raw[i] == miss_raw && tile < tiles && row < n) {  // bit equality with the missing pattern (DTPU.sv:653), before any transform
        r[i] = kQMissing;
        atomicOr(&tile_flags[tile], 1u);
      }
    }
    // within a feature row of the q tile, tuple t sits in dword (t % 512), half (t / 512): the 64 lanes of a scoring
    // wave then read 64 DIFFERENT dwords (two lanes sharing one dword at different byte addresses would 2-way
    // bank-conflict); here it makes one lane the owner of a whole dword
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
      const uint64_t tile = tile0 + 2u * (uint32_t)p2;
      if (tile < tiles) q32[(tile * W + j) * (uint64_t)(kQTile / 2) + t] = r[2 * p2] | (r[2 * p2 + 1] << 16);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Fused pre-pass for SMALL tables (all W tables + their bucket starts fit one CU's LDS, e.g. a 125-tree shard:
// ~1000 keys per feature): one kernel reads the tuples row-wise and writes the rank tiles, no transposed fp32
// intermediate -- HBM traffic 4F + 2F bytes per tuple instead of 4F + 4F + 4F + 2F.  The host packs the exact
// LDS image (ddt_image.cpp build_image_q16): per feature a skewed key table with >= P INT_MAX pads, 256 bucket
// starts (u16) and 8 parameter words {K, lo, shift, table byte offset, starts byte offset, 0, 0, 0}.
// A lane owns tuples t and t+512 of a tile (the two halves of one dword of the rank tile): quad-coalesced
// loads + DPP transpose as in score_tile_kernel, then per feature two searches and one 4-byte store.
// Work unit = 64 lane pairs (128 tuples) of one tile, handed out to WAVES through a global atomic counter: no
// barrier after the image load, the 16 waves of the block drift apart and cover each other's load latency, and
// blocks that start late (CUs busy with another stream's kernels, e.g. RCCL) simply take fewer units.
// ---------------------------------------------------------------------------------------------------
// One tuple line (4 features) of two rows against the tables resident in LDS: 8 searches advance together (the
// dependent LDS reads of one search are latency bound).  `par` = LDS byte address of the line's first parameter
// block {K, lo, span, table byte offset | starts byte offset, segment table byte offset, segment shift, 0} (layout and
// the segmented bucket index: ddt_image.cpp build_prepass_group).  Search = segment lookup (a 32-entry table: few distinct
// addresses per wave) + bucket start + log2(P) probes; keys past the bucket are > x by construction, no end test.
// Returns r(row0) | r(row1) << 16 per feature; in0 / in1 = the row exists (a missing value only counts there).
// Round 6 (counters: profiles/r06_prepass_valu.md -- these kernels were VALU-bound, 90-97 % of the issue slots, at ~57 instructions per value): the
// search position is carried as the BYTE ADDRESS of its table entry, so a probe's address is that register plus a DS immediate ((step - 1) * 4 for
// the unrolled steps 32 .. 1) and a step costs compare + select + add; the tables are linear (no i + i/32 skew to compute: the probes start at a
// bucket's own first key, nothing lines them up on one bank); IEEE is a template parameter (the key transform of cmp_mode 1 is not selected away
// per value in cmp_mode 0).
template <bool IEEE>
__device__ __forceinline__ u32x4 rank_line(const uint32_t par, const uint32_t P, const uint32_t miss_raw, const bool in0,
                                           const bool in1, const u32x4& v0, const u32x4& v1, bool& any_missing) {
  uint32_t K[4], tab[4], pos[4][2], missing = 0u;
  int32_t x[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint4 pa = lds_u4(par + (uint32_t)c * 32u);        // same address in every lane: {K, lo, span, table_off}
    const uint4 pb = lds_u4(par + (uint32_t)c * 32u + 16u);  // {starts_off, seg_off, seg_shift, 0}
    K[c] = pa.x;
    tab[c] = pa.w;
    const uint32_t seg_mask = (1u << pb.z) - 1u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t raw = h ? v1[c] : v0[c];
      if (raw == miss_raw && (h ? in1 : in0)) missing |= 1u << (2 * c + h);  // DTPU.sv:653, bit equality before any transform
      x[c][h] = (int32_t)(IEEE ? ieee_key(raw) : raw);
      uint32_t d = (uint32_t)x[c][h] - pa.y;
      d = x[c][h] < (int32_t)pa.y ? 0u : d;  // below the table: bucket 0, nothing there is <= x
      d = d < pa.z ? d : pa.z;               // above it: the last bucket, everything from there on is <= x
      const uint32_t sg = lds_u32(pb.y + (d >> pb.z) * 4u);  // first bucket | log2(bucket width) << 16
      const uint32_t bk = (sg & 0xFFFFu) + ((d & seg_mask) >> (sg >> 16));
      pos[c][h] = tab[c] + 4u * (uint32_t)*reinterpret_cast<const DDT_LDS(uint16_t)*>(pb.x + bk * 2u);  // byte address of the bucket's first key
    }
  }
  for (uint32_t step = P >> 1; step >= 64u; step >>= 1) {  // (buckets of 128 keys and more: degenerate key distributions only)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if ((int32_t)lds_u32(pos[c][h] + (step - 1u) * 4u) <= x[c][h]) pos[c][h] += step * 4u;  // probe < K + P: inside the padded table
    }
  }
#pragma unroll
  for (uint32_t step = 32u; step >= 1u; step >>= 1) {
    if (step < P) {  // wave-uniform
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if ((int32_t)lds_u32(pos[c][h] + (step - 1u) * 4u) <= x[c][h]) pos[c][h] += step * 4u;
      }
    }
  }
  any_missing = any_missing || missing != 0u;
  u32x4 out;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    uint32_t r[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      r[h] = (pos[c][h] - tab[c]) >> 2;
      r[h] = r[h] < K[c] ? r[h] : K[c];
      r[h] = (missing >> (2 * c + h)) & 1u ? kQMissing : r[h];
    }
    out[c] = r[0] | (r[1] << 16);
  }
  return out;
}

constexpr int kFusedThreads = 1024;  // two tiles per block and pass; 16 waves x 8 independent searches hide the LDS latency

// GRAB = units (64 lanes x 2 rows) a wave takes per ticket: 4 for the batches the throughput figures are taken on (one ticket per unit made 781 k
// same-address atomics per 100 M rows the bottleneck), 1 for a batch of a few tiles -- at 4 a single tile keeps TWO of a block's 16 waves busy, eight
// load -> rank rounds each: 21 us of a 47 us call (profiles/r06_small_batches.md)
template <bool IEEE, int GRAB = 4>
__global__ __launch_bounds__(kFusedThreads) void fused_rank_kernel(const uint32_t* __restrict__ tuples, uint64_t n, uint64_t n_pad, uint32_t W,
                                                                   const uint4* __restrict__ img_base, const PrepassPlan pl, uint32_t parts,
                                                                   uint32_t miss_raw, uint32_t ieee, uint32_t* __restrict__ q32,
                                                                   uint32_t* __restrict__ tile_flags, unsigned long long* __restrict__ counters, const uint32_t nt) {
  const uint32_t tid = threadIdx.x, t4 = tid & 3u, lpt = W / 4u;
  // one group (all tables resident): every block the same image, parts = 1.  Two groups of 4 lines: block b works for
  // group (b / parts) % 2 on row partition b % parts (see grouped_rank_kernel for why the partitions follow the XCDs)
  const uint32_t part = blockIdx.x % parts, g_own = (blockIdx.x / parts) % pl.groups;
  const uint64_t tiles_all = n_pad / kQTile;
  const uint64_t tiles = tiles_all > part ? (tiles_all - part + parts - 1u) / parts : 0u;  // of this partition
  if (tiles == 0u) return;
  const uint32_t img_bytes = pl.bytes[g_own], par_off = pl.par_off[g_own], P = pl.P[g_own];
  const uint32_t line_lo = pl.line_lo[g_own], line_hi = line_lo + pl.lines;
  const uint4* __restrict__ lds_img = img_base + pl.img_off[g_own] / 16u;
  unsigned long long* __restrict__ work_counter = counters + g_own * parts + part;
  for (uint32_t off = tid * 16u; off < img_bytes; off += kFusedThreads * 16u) lds_st_u4(off, lds_img[off / 16u]);
  __syncthreads();

  // one half-unit = lines 4g..4g+3 (16 features) of rows lt and lt+512: 8 x 16-byte loads per lane
  auto load_half = [&](u32x4 (&v)[2][4], uint64_t tile, uint32_t lt, uint32_t g) {
    const uint32_t line = 4u * g + t4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // load j: the 16 quads of the wave read 16 CONSECUTIVE rows (16j + quad) -- a linear 2 KiB span per instruction;
      // after the transposes lane (quad q, t) therefore owns row 16t + q of its 64-row slice (`own` below)
      const uint64_t slice_row = tile * kQTile + 512u * (uint32_t)h + (lt & ~63u) + ((lt >> 2) & 15u);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint64_t rj = slice_row + 16u * (uint32_t)j;
        const u32x4* src = reinterpret_cast<const u32x4*>(tuples + rj * W + 4u * line);
        v[h][j] = (rj < n && line < lpt) ? ((nt & 2u) ? __builtin_nontemporal_load(src) : *src) : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  // rank the 16 features of a half-tile; returns "this lane saw a missing value"
  auto rank_half = [&](u32x4 (&v)[2][4], uint64_t tile, uint32_t lt, uint32_t g) -> bool {
    bool any_missing = false;
    const uint32_t own = (lt & ~63u) + 16u * t4 + ((lt >> 2) & 15u);  // a permutation of the wave's 64 tuples
    quad_transpose(v[0], t4);  // v[h][i] = line 4g+i of row (tile*1024 + own + 512h)
    quad_transpose(v[1], t4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t line = 4u * g + (uint32_t)i;
      if (line >= lpt || line < line_lo || line >= line_hi) continue;  // this launch's feature group only
      const u32x4 r = rank_line<IEEE>(par_off + 4u * (line - line_lo) * 32u, P, miss_raw, tile * kQTile + own < n,
                                tile * kQTile + own + 512u < n, v[0][i], v[1][i], any_missing);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        uint32_t* dst = q32 + (tile * W + 4u * line + (uint32_t)c) * (uint64_t)(kQTile / 2) + own;
        if (nt & 1u) __builtin_nontemporal_store(r[c], dst);
        else *dst = r[c];
      }
    }
    return any_missing;
  };

  // (a software pipeline over half-tiles -- next half's loads in flight while this one is ranked -- measured
  // slower, 21.45 vs 20.58 ms end to end at 125 trees: at 128 VGPRs it spills; 16 waves hide the latency well enough)
  // a wave takes 4 units (half a tile) per grab: one grab per unit made 781 k same-address atomics per 100 M rows
  // the bottleneck (measured 9.3 instead of 5.4 ms)
  const unsigned long long units = (unsigned long long)tiles * 8u;
  for (;;) {
    unsigned long long u0 = 0;
    if ((tid & 63u) == 0u) u0 = atomicAdd(work_counter, (unsigned long long)GRAB);
    u0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u0 >> 32)) << 32) |
         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u0);
    if (u0 >= units) break;
    const uint64_t tile = (u0 >> 3) * parts + part;
    bool miss = false;
    for (uint32_t k = 0; k < (uint32_t)GRAB; ++k) {
      const uint32_t lt = ((((uint32_t)u0 & 7u) + k) << 6) | (tid & 63u);
#pragma unroll
      for (uint32_t g = 0; g < 2u; ++g) {
        if (4u * g >= line_hi || 4u * g + 4u <= line_lo) continue;  // half-row without lines of this group (wave-uniform)
        u32x4 v[2][4];
        load_half(v, tile, lt, g);
        miss |= rank_half(v, tile, lt, g);
      }
    }
    if (miss) atomicOr(&tile_flags[tile], 1u);
  }
}

// ---------------------------------------------------------------------------------------------------
// Grouped pre-pass for BIG tables (1000 trees x 255 nodes over 32 features: ~8 k keys = 33 KiB per feature, 1 MiB in all):
// no transposed fp32 intermediate either.  The features are cut into G = 4 or 8 groups of L = 2 or 1 tuple lines whose
// tables fit one CU's LDS; ONE launch, block b works for group (b / parts) % G on the row partition b % parts
// (workgroups go to the XCDs round-robin, so parts = 8 puts the blocks of all G groups that read the same rows behind the
// same L2: a row is fetched from HBM once and the other G - 1 reads hit that L2 / the Infinity Cache).  HBM traffic
// 4F + 2F bytes per tuple instead of 4F + 4F + 4F + 2F, and no [W][n] fp32 workspace.  A lane owns rows t and t+512 of
// a tile and reads ITS line(s) of them directly (16-byte loads at a 128-byte stride; the neighbours in the row belong
// to other groups), ranks 4 features x 2 rows in lock-step (rank_line) and stores one fully coalesced dword per feature.
// Work = half tiles, handed to waves through one atomic counter per (group, part); the loads of the next 64 lane
// pairs are in flight while the current ones are ranked.  (Deeper prefetch -- whole-tile grabs with 8 x 16-byte loads per
// lane in flight -- measured SLOWER, 7.3 vs 5.7 ms at 1000 trees: every 16-byte load pulls a whole line into the L2 the
// groups share, and the footprint in flight then exceeds it.)
// ---------------------------------------------------------------------------------------------------
template <int L, bool IEEE, int GRAB = 4>  // GRAB: fused_rank_kernel
__global__ __launch_bounds__(kFusedThreads) void grouped_rank_kernel(const uint32_t* __restrict__ tuples, uint64_t n, uint64_t n_pad, uint32_t W,
                                                                     const uint4* __restrict__ img_base, const PrepassPlan pl, uint32_t parts,
                                                                     uint32_t miss_raw, uint32_t ieee, uint32_t* __restrict__ q32,
                                                                     uint32_t* __restrict__ tile_flags, unsigned long long* __restrict__ counters, const uint32_t nt) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, lpt = W / 4u;
  const uint32_t part = blockIdx.x % parts, g = (blockIdx.x / parts) % pl.groups;
  const uint64_t tiles = n_pad / kQTile;
  const uint64_t tiles_part = tiles > part ? (tiles - part + parts - 1u) / parts : 0u;
  if (tiles_part == 0u) return;
  const uint32_t img_bytes = pl.bytes[g], par_off = pl.par_off[g], P = pl.P[g], line_lo = pl.line_lo[g];
  const uint4* __restrict__ img = img_base + pl.img_off[g] / 16u;
  for (uint32_t off = tid * 16u; off < img_bytes; off += kFusedThreads * 16u) lds_st_u4(off, img[off / 16u]);
  __syncthreads();
  unsigned long long* __restrict__ counter = counters + g * parts + part;
  const unsigned long long units = (unsigned long long)tiles_part * 8u;

  // L = 1: a lane reads its own two rows' line (16 bytes of every 128-byte row: the rest belongs to the other groups).
  // L = 2: the two lanes of a pair read the 32 contiguous bytes of ONE row (lane parity = line), 32 consecutive rows per
  // instruction, then swap halves (DPP): lane (pair p, parity t) ends up with both lines of row 32t + p of the wave's
  // 64-row slice (`own`).  Half of every 64-byte sector pulled through the L1 is used instead of a quarter.
  const uint32_t t2 = lane & 1u;
  auto load_unit = [&](u32x4 (&v)[L][2], uint64_t tile, uint32_t lt) {
#pragma unroll
    for (int l = 0; l < L; ++l) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint64_t row = tile * kQTile + 512u * (uint32_t)h + (L == 2 ? (lt & ~63u) + 32u * (uint32_t)l + ((lt >> 1) & 31u) : lt);
        const uint32_t line = line_lo + (L == 2 ? t2 : (uint32_t)l);
        const u32x4* src = reinterpret_cast<const u32x4*>(tuples + row * W + 4u * line);
        v[l][h] = (row < n && line < lpt) ? ((nt & 2u) ? __builtin_nontemporal_load(src) : *src) : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto pair_swap = [&](u32x4 (&v)[L][2]) {  // L == 2: v[l][h] = line (line_lo + l) of the lane's own row
    if (L == 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t x = v[0][h][c], y = v[L - 1][h][c];
          const uint32_t px = dpp_xor1(x), py = dpp_xor1(y);  // cross-lane reads with every lane active, THEN select
          v[0][h][c] = t2 ? py : x;
          v[L - 1][h][c] = t2 ? y : px;
        }
      }
    }
  };

  for (;;) {
    unsigned long long u0 = 0;
    if (lane == 0u) u0 = atomicAdd(counter, (unsigned long long)GRAB);
    u0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u0 >> 32)) << 32) |
         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u0);
    if (u0 >= units) break;
    const uint64_t tile = (u0 >> 3) * parts + part;
    const uint32_t lt0 = (((uint32_t)u0 & 7u) << 6) | lane;  // the GRAB units of this grab: lt0, lt0 + 64, .. (< 512)
    bool miss = false;
    u32x4 cur[L][2], nxt[L][2];
    load_unit(cur, tile, lt0);
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)GRAB; ++k) {
      const uint32_t lt = lt0 + 64u * k;
      if (k + 1u < (uint32_t)GRAB) load_unit(nxt, tile, lt + 64u);
      pair_swap(cur);
      const uint32_t own = L == 2 ? (lt & ~63u) + 32u * t2 + ((lt >> 1) & 31u) : lt;  // a permutation of the wave's 64 tuples
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const uint32_t line = line_lo + (uint32_t)l;
        if (line >= lpt) continue;  // narrow tuples: the last group may be short (wave-uniform)
        const u32x4 r = rank_line<IEEE>(par_off + 4u * (uint32_t)l * 32u, P, miss_raw, tile * kQTile + own < n, tile * kQTile + own + 512u < n,
                                  cur[l][0], cur[l][1], miss);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          uint32_t* dst = q32 + (tile * W + 4u * line + (uint32_t)c) * (uint64_t)(kQTile / 2) + own;
          if (nt & 1u) __builtin_nontemporal_store(r[c], dst);
          else *dst = r[c];
        }
      }
      if (k + 1u < (uint32_t)GRAB) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
          cur[l][0] = nxt[l][0];
          cur[l][1] = nxt[l][1];
        }
      }
    }
    if (miss) atomicOr(&tile_flags[tile], 1u);
  }
}

// the rank pre-pass of one batch: q tiles + per-tile missing flags into the workspace of `x` (shared by the perfect-tree q16
// kernels and the rank-quantised sparse kernels, ddt_sparse.hip)
hipError_t launch_q16_prepass(const ScoreArgs& a, const Q16Aux& x, hipStream_t s) {
  const uint64_t tiles = (a.n + kQTile - 1) / kQTile;
  if (tiles == 0) return hipSuccess;
  const uint32_t W = a.tuple_words;
  const uint32_t rank_lds = x.Kpad * 4u + kRankBuckets * 2u;  // table + bucket starts, see rank_kernel
  auto rk = a.ieee ? rank_kernel<true> : rank_kernel<false>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rank_lds);
  if (e != hipSuccess) return e;
  // tile flags + (8-byte aligned, right behind them) the work counters of the fused / grouped pre-pass
  unsigned long long* counter = reinterpret_cast<unsigned long long*>(x.tile_flags + ((tiles + 1u) & ~(uint64_t)1u));
  e = launch_zero_words(x.tile_flags, ((tiles + 1u) & ~(uint64_t)1u) + 2u * kQ16GroupedCounters + kQ16TileCounterWords, s);  // + the _p kernels' tile counter
  if (e != hipSuccess) return e;
  const PrepassPlan& pp = x.prepass;
  // a batch of a few tiles: a unit per ticket, blocks for every two tiles (option-free: the result is the same bits, only who ranks what changes)
  static const bool small_on = [] {  // A/B: DDT_PREPASS_SMALL=0 -> four units per ticket whatever the batch
    const char* v = getenv("DDT_PREPASS_SMALL");
    return !(v && v[0] == '0');
  }();
  // (measured, profiles/r06_small_batches.md: the fused form wins up to 64 tiles -- 100 x d6: 54.5 against 68.1 us per call there --, the grouped form up
  // to 16 -- 1000 x d8 at 64 tiles: 131.9 against 122.1)
  const bool small = small_on && tiles <= ((pp.groups && pp.lines >= 4u) ? 64u : 16u);
  if (pp.groups && pp.lines >= 4u) {
    // 1 group (all tables fit one CU's LDS together) or 2 groups of 4 lines: quad-coalesced loads, every 64-byte sector a block
    // pulls is used whole (fused_rank_kernel); at most one block per CU
    uint32_t lds = 0;
    for (uint32_t g = 0; g < pp.groups; ++g) lds = pp.bytes[g] > lds ? pp.bytes[g] : lds;
    const uint32_t parts = (pp.groups > 1u && a.num_cus >= 8u * pp.groups && tiles >= 8u) ? 8u : 1u;
    uint32_t per_pair = a.num_cus / (parts * pp.groups);
    if (per_pair < 1u) per_pair = 1u;
    const uint64_t tiles_part = (tiles + parts - 1u) / parts;  // a block's 16 waves take 8 tiles per round (small batches: 2, a unit per wave)
    const uint64_t per_block = small ? 2u : 8u;
    if ((uint64_t)per_pair > (tiles_part + per_block - 1u) / per_block) per_pair = (uint32_t)((tiles_part + per_block - 1u) / per_block);
    auto fk = small ? (a.ieee ? fused_rank_kernel<true, 1> : fused_rank_kernel<false, 1>) : (a.ieee ? fused_rank_kernel<true> : fused_rank_kernel<false>);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fk, dim3(parts * pp.groups * per_pair), dim3(kFusedThreads), lds, s, a.tuples, a.n, x.n_pad, W, x.prepass_img, pp,
                       parts, a.miss_raw, a.ieee, reinterpret_cast<uint32_t*>(x.q), x.tile_flags, counter, x.prepass_nt);
  } else if (pp.groups) {  // one launch, blocks split over feature groups x row partitions (XCDs)
    uint32_t lds = 0;
    for (uint32_t g = 0; g < pp.groups; ++g) lds = pp.bytes[g] > lds ? pp.bytes[g] : lds;
    // one block per CU (the image takes most of the LDS); 8 row partitions when every (group, partition) gets a block
    const uint32_t parts = (a.num_cus >= 8u * pp.groups && tiles >= 8u) ? 8u : 1u;
    uint32_t per_pair = a.num_cus / (parts * pp.groups);
    if (per_pair < 1u) per_pair = 1u;
    const uint64_t tiles_part = (tiles + parts - 1u) / parts;  // a block's 16 waves take 8 tiles per round (small batches: 2, a unit per wave)
    const uint64_t per_block = small ? 2u : 8u;
    if ((uint64_t)per_pair > (tiles_part + per_block - 1u) / per_block) per_pair = (uint32_t)((tiles_part + per_block - 1u) / per_block);
    const uint32_t grid = parts * pp.groups * per_pair;
    auto gk = small ? (pp.lines == 1u ? (a.ieee ? grouped_rank_kernel<1, true, 1> : grouped_rank_kernel<1, false, 1>)
                                      : (a.ieee ? grouped_rank_kernel<2, true, 1> : grouped_rank_kernel<2, false, 1>))
                    : (pp.lines == 1u ? (a.ieee ? grouped_rank_kernel<1, true> : grouped_rank_kernel<1, false>)
                                      : (a.ieee ? grouped_rank_kernel<2, true> : grouped_rank_kernel<2, false>));
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gk, dim3(grid), dim3(kFusedThreads), lds, s, a.tuples, a.n, x.n_pad, W, x.prepass_img, pp, parts, a.miss_raw, a.ieee,
                       reinterpret_cast<uint32_t*>(x.q), x.tile_flags, counter, x.prepass_nt);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(transpose_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((W + 1) * 256 * 4));
    if (e != hipSuccess) return e;
    if (!x.skip_transpose) {
      if (x.fmap) {  // feature compaction: only the columns the model tests, out of rows of in_words words
        e = launch_gather_transpose(a.tuples, x.in_words, x.fmap, W, a.n, x.n_pad, x.xT, s);
        if (e != hipSuccess) return e;
      } else {
        hipLaunchKernelGGL(transpose_kernel, dim3((uint32_t)(x.n_pad / 256)), dim3(256), (W + 1) * 256 * 4, s, a.tuples, W, a.n, x.n_pad, x.xT);
      }
    }
    // grid-stride over tiles; blockIdx.y = feature; the table (up to 152 KiB) is loaded once per block, so the blocks are as few and as
    // long-lived as fill the chip: resident blocks per CU (one with a big table, two when two fit) x CUs, split over the features.  (Until
    // round 5: up to 512 blocks per feature -- 16384 blocks of ~19 tiles each at 32 features, a third of whose time was the table load:
    // 1.19 ms per 10 M tuples x 32 features with 32 k keys each; DDT_RANK_GRID_OLD=1 brings that grid back for A/B.)
    uint32_t bx = (uint32_t)((x.n_pad / kQTile + 3) / 4);  // 4 tiles (kRankThreads x 4 rows) per block and pass
    static const bool old_grid = [] {
      const char* v = getenv("DDT_RANK_GRID_OLD");
      return v && v[0] && v[0] != '0';
    }();
    if (old_grid) {
      if (bx > 512u) bx = 512u;
    } else {
      const uint32_t per_cu = rank_lds <= 80u * 1024u ? 2u : 1u;
      const uint32_t want = (per_cu * a.num_cus + W - 1u) / W;
      if (bx > want) bx = want < 1u ? 1u : want;
    }
    hipLaunchKernelGGL(rk, dim3(bx, W), dim3(kRankThreads), rank_lds, s, x.xT, a.n, x.n_pad, x.tables, x.Kpad, x.tabP, x.tabS, a.miss_raw, W, x.q,
                       x.tile_flags);
  }
  return hipGetLastError();
}
}  // namespace ddt
