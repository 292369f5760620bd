// ddt_device.h -- device-side helpers shared by the kernel translation units (ddt_kernels.hip, ddt_sparse.hip):
// absolute-address LDS access, the reference-order accumulator ring, the compare rule, the perfect-tree walk,
// model-chunk staging (registers / global->LDS DMA), tuple staging helpers and the in-quad DPP transpose.
#pragma once
#include <hip/hip_runtime.h>

#include "ddt_internal.h"

namespace ddt {

// ---------------------------------------------------------------------------------------------------
// LDS access by ABSOLUTE byte address.  The kernels declare no static __shared__ object, so the dynamic
// LDS segment starts at address 0 (tests check group_segment_fixed_size == 0) and a DS address is just
// the byte offset.  Going through the `extern __shared__` symbol instead makes hipcc add the (link-time)
// symbol address to every DS address -- one wasted VALU op per node visit.
// ---------------------------------------------------------------------------------------------------
#define DDT_LDS(T) __attribute__((address_space(3))) T
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_u2(uint32_t a) {
  const u32x2 v = *reinterpret_cast<const DDT_LDS(u32x2)*>(a);
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint4 lds_u4(uint32_t a) {
  const u32x4 v = *reinterpret_cast<const DDT_LDS(u32x4)*>(a);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t lds_u32(uint32_t a) { return *reinterpret_cast<const DDT_LDS(uint32_t)*>(a); }
__device__ __forceinline__ float lds_f32(uint32_t a) { return *reinterpret_cast<const DDT_LDS(float)*>(a); }
__device__ __forceinline__ void lds_st_u32(uint32_t a, uint32_t v) { *reinterpret_cast<DDT_LDS(uint32_t)*>(a) = v; }
__device__ __forceinline__ void lds_st_u4(uint32_t a, uint4 v) {
  u32x4 t = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<DDT_LDS(u32x4)*>(a) = t;
}

// ---------------------------------------------------------------------------------------------------
// The reference's fp32 adder vs IEEE-754 (sum_mode 2, include/ddt.h).  On the leaf domain the loader enforces (+0 or
// normal, 2^-102 <= |leaf| < 2^96: no partial sum can be sub-normal, overflow or be -0) the FloPoCo adder of
// rtl/DTEngine/common/FPAdder_2cycles_latency.v IS the IEEE round-to-nearest-even add, with ONE exception
// (:325-326, `shiftedOut = (expDiff >= 25)` forces the alignment shift to 26 already at an exponent difference of 25): an
// effective subtraction whose larger operand is an exact power of two, exponents 25 apart, smaller mantissa != 0.  IEEE
// rounds to the float just below the power of two; the RTL returns the power of two unchanged.
// sum_suspect(): a cheap NECESSARY condition on the IEEE result (its mantissa is all ones there; the low 16 bits are tested:
// one v_cmp_eq_u16 per add); radd_exact(): the reference adder's result.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sum_suspect(float s) { return (uint16_t)__float_as_uint(s) == (uint16_t)0xFFFFu; }
__device__ __forceinline__ float radd_exact(float a, float b) { return ref_add_exact(a, b); }
template <bool EX>
__device__ __forceinline__ float radd(float a, float b) { return EX ? radd_exact(a, b) : a + b; }

// ---------------------------------------------------------------------------------------------------
// reference-order fp32 accumulation state (per lane, per tuple)
//   tree i -> PU i%8; group g=i/8 -> cluster g%C; per cluster acc <- s_g + acc in slot order; final
//   sequential add over clusters.  SURVEY.md 8(a) A4/A11/A12.
// The C cluster accumulators live in registers a[0..C-1] and are ROTATED after every group so that the
// current cluster is always a[0]: every index below is a compile-time constant.  (A `switch (cluster)`
// over acc[k] gets merged by hipcc into one dynamically indexed access, which lands in scratch.)
// ---------------------------------------------------------------------------------------------------
template <int R>
struct RefAcc {
  float a[R][8];
  float half[R];
  uint32_t pos;  // wave-uniform: groups pushed so far, mod C
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      half[r] = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) a[r][k] = 0.f;
    }
    pos = 0;
  }
  __device__ __forceinline__ void rotate(uint32_t C) {  // a[k] <- a[k+1], a[C-1] <- a[0]
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float t = a[r][0];
      if (C == 8u) {
        a[r][0] = a[r][1]; a[r][1] = a[r][2]; a[r][2] = a[r][3]; a[r][3] = a[r][4];
        a[r][4] = a[r][5]; a[r][5] = a[r][6]; a[r][6] = a[r][7]; a[r][7] = t;
      } else if (C == 4u) {
        a[r][0] = a[r][1]; a[r][1] = a[r][2]; a[r][2] = a[r][3]; a[r][3] = t;
      } else if (C == 2u) {
        a[r][0] = a[r][1]; a[r][1] = t;
      }
    }
  }
  // the current cluster's accumulator takes its new value (s_g + acc, FPAggregator.v:124-131) and the ring moves on
  __device__ __forceinline__ void commit_group(const float (&acc_new)[R], uint32_t C) {
#pragma unroll
    for (int r = 0; r < R; ++r) a[r][0] = acc_new[r];
    rotate(C);
    pos = (pos + 1u == C) ? 0u : pos + 1u;
  }
  // finish the current turn of the ring so that a[k] is cluster k again
  __device__ __forceinline__ void align(uint32_t C) {
    while (pos != 0u) {
      rotate(C);
      pos = (pos + 1u == C) ? 0u : pos + 1u;
    }
  }
  // sequential add over clusters c = 0..C-1 (Core.sv:486-541); call align() first.  `exact`: sum_mode 2
  __device__ __forceinline__ float total(int r, uint32_t C, bool exact = false) const {
    float t = 0.f;
    if (!exact) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if ((uint32_t)k < C) t = a[r][k] + t;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if ((uint32_t)k < C) t = radd_exact(a[r][k], t);
    }
    return t;
  }
  // the same sum WITHOUT align(): ring slot k holds cluster (pos + k) mod C, so cluster c sits in slot (c - pos) mod C; one
  // wave-uniform branch per possible pos instead of up to C - 1 rotations of 8 moves (the stream kernel's 56 v_mov per
  // tuple).  sum_mode 2 keeps the align() route (its adds are long; fifteen inlined orders of them would only grow the code).
  template <int CC, int P>
  __device__ __forceinline__ float seq(int r) const {
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < CC; ++c) t = a[r][(c - P + CC) % CC] + t;
    return t;
  }
  template <int CC, int P = 0>
  __device__ __forceinline__ float by_pos(int r) const {
    if constexpr (P == CC - 1) {
      return seq<CC, P>(r);
    } else {
      if (pos == (uint32_t)P) return seq<CC, P>(r);
      return by_pos<CC, P + 1>(r);
    }
  }
  __device__ __forceinline__ float total_ring(int r, uint32_t C, bool exact = false) {
    if (exact) {
      align(C);
      return total(r, C, true);
    }
    if (C == 8u) return by_pos<8>(r);
    if (C == 4u) return by_pos<4>(r);
    if (C == 2u) return by_pos<2>(r);
    return seq<1, 0>(r);
  }
};

// fold the leaves of one sub-group of U trees (stream order) into the accumulators.  SUM 1: fp64 in stream order.
// SUM 0: the reference's adder network -- FPAddersReduceTree.sv:94-141 ((l0+l1)+(l2+l3))+((l4+l5)+(l6+l7)) per PU group,
// then acc <- s + acc on the group's cluster; with IEEE adds (sum_mode 0), or `exact` (sum_mode 2) = the reference adder
// itself: the IEEE adds run as always, every result is tested with sum_suspect(), and only a wave in which some lane
// trips the test recomputes the group with radd_exact() (a wave-uniform branch around the rare path).
// maximum of the low 16 bits of three registers (inline asm: written as a C++ maximum of uint16_t, hipcc emits v_cmp_gt_u32_sdwa +
// v_cndmask per operand)
__device__ __forceinline__ uint32_t max3_lo16(uint32_t p, uint32_t q, uint32_t r) {
  uint32_t d;
  asm("v_max3_u16 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(q), "v"(r));
  return d;
}
__device__ __forceinline__ uint32_t fbits(float v) { return __float_as_uint(v); }

// EX (sum_mode 2): every IEEE sum of the fold has to pass sum_suspect().  Round 4: not one v_cmp_eq_u16 per sum, but the running
// MAXIMUM of the sums' low 16 bits -- 0xFFFF iff some sum is suspect -- taken two sums per v_max3_u16, and ONE test per fold.
template <int U, int R, bool EX>
__device__ __forceinline__ void fold_ref(const float (&lf)[R][U], const int phase, const uint32_t C, RefAcc<R>& ra) {
  uint32_t mx = 0u;  // EX only
  auto any_suspect = [&]() -> bool { return __ballot((uint16_t)mx == (uint16_t)0xFFFFu) != 0ull; };  // wave-uniform
  if (U == 8) {
    float acc_new[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float s1 = lf[r][0] + lf[r][1], s2 = lf[r][2] + lf[r][3], s3 = s1 + s2;
      const float s4 = lf[r][4 % U] + lf[r][5 % U], s5 = lf[r][6 % U] + lf[r][7 % U], s6 = s4 + s5, s7 = s3 + s6;
      acc_new[r] = s7 + ra.a[r][0];
      if (EX) {
        mx = r == 0 ? max3_lo16(fbits(s1), fbits(s2), fbits(s3)) : max3_lo16(mx, fbits(s1), fbits(s2));
        if (r != 0) mx = max3_lo16(mx, fbits(s3), fbits(s3));
        mx = max3_lo16(mx, fbits(s4), fbits(s5));
        mx = max3_lo16(mx, fbits(s6), fbits(s7));
        mx = max3_lo16(mx, fbits(acc_new[r]), fbits(acc_new[r]));
      }
    }
    if (EX && any_suspect()) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float e = radd_exact(radd_exact(radd_exact(lf[r][0], lf[r][1]), radd_exact(lf[r][2], lf[r][3])),
                                   radd_exact(radd_exact(lf[r][4 % U], lf[r][5 % U]), radd_exact(lf[r][6 % U], lf[r][7 % U])));
        acc_new[r] = radd_exact(e, ra.a[r][0]);
      }
    }
    ra.commit_group(acc_new, C);
  } else if (phase == 0) {
    float p[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float s1 = lf[r][0] + lf[r][1], s2 = lf[r][2] + lf[r][3];
      p[r] = s1 + s2;
      if (EX) {
        mx = r == 0 ? max3_lo16(fbits(s1), fbits(s2), fbits(p[r])) : max3_lo16(mx, fbits(s1), fbits(s2));
        if (r != 0) mx = max3_lo16(mx, fbits(p[r]), fbits(p[r]));
      }
    }
    if (EX && any_suspect()) {
#pragma unroll
      for (int r = 0; r < R; ++r) p[r] = radd_exact(radd_exact(lf[r][0], lf[r][1]), radd_exact(lf[r][2], lf[r][3]));
    }
#pragma unroll
    for (int r = 0; r < R; ++r) ra.half[r] = p[r];
  } else {
    float acc_new[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float s1 = lf[r][0] + lf[r][1], s2 = lf[r][2] + lf[r][3], s3 = s1 + s2, s4 = ra.half[r] + s3;
      acc_new[r] = s4 + ra.a[r][0];
      if (EX) {
        mx = r == 0 ? max3_lo16(fbits(s1), fbits(s2), fbits(s3)) : max3_lo16(mx, fbits(s1), fbits(s2));
        if (r != 0) mx = max3_lo16(mx, fbits(s3), fbits(s3));
        mx = max3_lo16(mx, fbits(s4), fbits(acc_new[r]));
      }
    }
    if (EX && any_suspect()) {
#pragma unroll
      for (int r = 0; r < R; ++r)
        acc_new[r] = radd_exact(radd_exact(ra.half[r], radd_exact(radd_exact(lf[r][0], lf[r][1]), radd_exact(lf[r][2], lf[r][3]))), ra.a[r][0]);
    }
    ra.commit_group(acc_new, C);
  }
}

template <int U, int R, int SUM>
__device__ __forceinline__ void fold_leaves(const float (&lf)[R][U], const int phase /*U==4: 0 first half, 1 second*/,
                                            const uint32_t C, RefAcc<R>& ra, double (&dacc)[R], const bool exact = false) {
  static_assert(U == 4 || U == 8, "sub-group = half a PU group or a whole one");
  if (SUM == 1) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < R; ++r) dacc[r] += (double)lf[r][u];  // stream order, fp64
  } else if (!exact) {
    fold_ref<U, R, false>(lf, phase, C, ra);
  } else {
    fold_ref<U, R, true>(lf, phase, C, ra);
  }
}

// ---------------------------------------------------------------------------------------------------
// walk U trees x R tuples through D levels; returns the selected leaves.
//   `base` = LDS byte address of the first tree (compile-time in the tile kernel => DS immediates; a
//   wave-uniform runtime value in the stream kernel), trees TREE_BYTES apart.
//   FUSED (image layout 1): levels 0..D-2 are 8-byte records of the 1-based heap at byte 8*m; the last
//   level is 16-byte records {thr, w2, leafL, leafR} at 4*2^D + 16*(m - 2^(D-1)), so the leaf comes
//   with its parent (one LDS round trip and one DS op less per tree).
// ---------------------------------------------------------------------------------------------------
// (x & 0xFFFFFF) + y in one VALU instruction (hipcc turns the multiply by one into v_and + v_add)
__device__ __forceinline__ uint32_t low24_plus(uint32_t x, uint32_t y) {
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, 1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

template <bool SLOW>
__device__ __forceinline__ bool go_right(uint32_t f, uint32_t thr, uint32_t w2, uint32_t miss_key) {
  bool right = (int32_t)f >= (int32_t)thr;                         // !(feature < threshold), DTPU.sv:655-657
  if (SLOW) right = (f == miss_key) ? ((w2 >> 31) != 0u) : right;  // DTPU.sv:653,667
  return right;
}

// ADD (stream kernel): the feature rows are not multiples of the row size apart (Variant::feat_word_stream), so the lane's
// column offset is ADDED to the low 24 bits of the node's feature word -- one v_mad_u32_u24 (x1), which also drops the
// miss_right flag in bit 31 -- instead of OR-ed (v_and_or_b32)
template <int D, int U, int R, int TREE_BYTES, bool SLOW, bool FUSED, bool ADD = false>
__device__ __forceinline__ void walk_trees(const uint32_t base, const uint32_t (&lane_off)[R], const uint32_t miss_key,
                                           float (&leaf)[R][U]) {
  constexpr int LAST = FUSED ? D - 1 : D;  // levels walked over 8-byte records
  uint32_t m8[R][U];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int u = 0; u < U; ++u) m8[r][u] = 8u;  // root of the 1-based heap, in bytes

#pragma unroll
  for (int lvl = 0; lvl < LAST; ++lvl) {
    uint2 nd[R][U];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u) nd[r][u] = lds_u2(m8[r][u] + (base + (uint32_t)(u * TREE_BYTES)));  // ds_read_b64
    uint32_t f[R][U];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u)  // ds_read_b32, conflict-free
        f[r][u] = lds_u32(ADD ? low24_plus(nd[r][u].y, lane_off[r]) : ((nd[r][u].y & 0x7FFFFFFFu) | lane_off[r]));
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u)
        m8[r][u] = (m8[r][u] << 1) + (go_right<SLOW>(f[r][u], nd[r][u].x, nd[r][u].y, miss_key) ? 8u : 0u);
  }
  static_assert(!(FUSED && ADD), "the fused last level is the tile kernel's");
  if (FUSED) {
    uint4 rec[R][U];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u)  // ds_read_b128; offset = tree base - 4*2^D >= 0 because of Variant::model_base()
        rec[r][u] = lds_u4((m8[r][u] << 1) + (base + (uint32_t)(u * TREE_BYTES + (4 << D) - (8 << D))));
    uint32_t f[R][U];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u) f[r][u] = lds_u32((rec[r][u].y & 0x7FFFFFFFu) | lane_off[r]);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u)
        leaf[r][u] = __uint_as_float(go_right<SLOW>(f[r][u], rec[r][u].x, rec[r][u].y, miss_key) ? rec[r][u].w : rec[r][u].z);
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u) leaf[r][u] = lds_f32((m8[r][u] >> 1) + (base + (uint32_t)(u * TREE_BYTES + (4 << D))));
  }
}

// ---------------------------------------------------------------------------------------------------
// model chunk staging (tile kernel)
// ---------------------------------------------------------------------------------------------------
template <int THREADS, int CHUNK_BYTES>
struct StageRegs {
  static constexpr int UNITS = CHUNK_BYTES / 16;
  static constexpr int PER = (UNITS + THREADS - 1) / THREADS;
  uint4 v[PER];
  __device__ __forceinline__ void load(const uint4* __restrict__ img, uint32_t k, int tid) {
    const uint4* src = img + (size_t)k * UNITS;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int unit = tid + i * THREADS;
      if (unit < UNITS) v[i] = src[unit];
    }
  }
  __device__ __forceinline__ void commit(int buf_off, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int unit = tid + i * THREADS;
      if (unit < UNITS) lds_st_u4((uint32_t)(buf_off + unit * 16), v[i]);
    }
  }
};

// global -> LDS DMA (global_load_lds_dwordx4): LDS destination = M0 (wave-uniform base) + lane*16.
// Issued through inline asm on purpose: with the builtin, hipcc (ROCm 7.2) cannot prove that the DMA's
// LDS write does not alias the ds_reads of the *other* buffer and puts `s_waitcnt vmcnt(0)` in front of
// the first ds_read of the compute phase, which serialises the prefetch with the compute.  The asm form
// is invisible to that pass; the kernel waits for it itself (vmcnt(0) right before the chunk barrier).
template <int THREADS, int CHUNK_BYTES>
__device__ __forceinline__ void dma_chunk(const uint4* __restrict__ img, uint32_t k, int buf_off, int tid) {
  constexpr int UNITS = CHUNK_BYTES / 16;
  static_assert(UNITS % 64 == 0, "whole waves per DMA");
  constexpr int PER = (UNITS + THREADS - 1) / THREADS;
  const uint4* src = img + (size_t)k * UNITS;
  const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (wave_base + i * THREADS < UNITS) {  // wave-uniform
      const uint4* g = src + (tid + i * THREADS);
      const uint32_t lds_addr = (uint32_t)(buf_off + (wave_base + i * THREADS) * 16);  // dynamic LDS starts at 0
      asm volatile(
          "s_mov_b32 m0, %0\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, off"
          :
          : "s"(lds_addr), "v"(g)
          : "memory");  // m0 is a reserved register for hipcc: it re-materialises m0 before its own uses
    }
  }
}

// raw / IEEE-key transform of one staged feature word + missing detection
__device__ __forceinline__ uint32_t stage_word(uint32_t v, const ScoreArgs& a, uint32_t& miss_any, bool valid) {
  const uint32_t m = (v == a.miss_raw) ? 1u : 0u;
  miss_any |= m & (valid ? 1u : 0u);
  if (a.ieee) v = m ? kMissSentinelIeee : ieee_key(v);  // wave-uniform branch
  return v;
}

// block-wide OR through one dynamic-LDS word per wave (no static __shared__, see lds_* above); contains the
// barrier that publishes the staged tile
template <int THREADS>
__device__ __forceinline__ bool block_any(uint32_t flag, uint32_t flags_addr, int tid) {
  const unsigned long long wave_any = __ballot(flag != 0u);
  if ((tid & 63) == 0) lds_st_u32(flags_addr + (uint32_t)(tid >> 6) * 4u, wave_any != 0ull ? 1u : 0u);
  __syncthreads();
  uint32_t any = 0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) any |= lds_u32(flags_addr + (uint32_t)w * 4u);
  return __builtin_amdgcn_readfirstlane(any) != 0u;
}

// 4x4 transpose inside a lane quad: in: lane t holds v[j] = element (row j, column t); out: v[i] = (row t, column i).
// Two butterfly stages (partner lane t^1, then t^2) through DPP quad_perm moves; no LDS traffic.
__device__ __forceinline__ uint32_t dpp_xor1(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true); }
__device__ __forceinline__ void quad_transpose(u32x4 (&v)[4], uint32_t t) {
  const bool odd = (t & 1u) != 0u, hi = (t & 2u) != 0u;
#pragma unroll
  for (int p = 0; p < 4; p += 2) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t x = v[p][c], y = v[p + 1][c];
      const uint32_t px = dpp_xor1(x), py = dpp_xor1(y);  // cross-lane reads with every lane active, THEN select
      v[p][c] = odd ? py : x;
      v[p + 1][c] = odd ? y : px;
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t x = v[p][c], z = v[p + 2][c];
      const uint32_t px = dpp_xor2(x), pz = dpp_xor2(z);
      v[p][c] = hi ? pz : x;
      v[p + 2][c] = hi ? z : px;
    }
  }
}

}  // namespace ddt
