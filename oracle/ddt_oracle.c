/*
 * ddt_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see ddt_oracle.h).
 *
 * PARITY: adder, compare rule, group tree, accumulator datapath, chain hop, the traversal datapath, the tree ->
 * cluster / PU schedule and the programming side (how the streams land in the PU memories, per-tree offsets, EMPTY
 * slots) are pinned against vectors evaluated / executed from the reference's own RTL source
 * (tests/golden/make_rtl_golden.py, make_schedule_golden.py, make_program_golden.py);
 * *** everything else is UNPINNED *** -- the reference (FPGA RTL) has no tests/golden vectors and its
 * handshake / FIFO control cannot be run here (see ddt_oracle.h).
 * This file restates the RTL's scoring semantics; each function cites the lines it follows.
 * All paths below are relative to /root/reference/rtl/DTEngine/.
 */
#define _GNU_SOURCE
#include "ddt_oracle.h"

#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* =============================================================================================
 * 1. The reference's fp32 adder: FloPoCo FPAdder_8_23 (common/FPAdder_2cycles_latency.v:210-387)
 *    modelled at bit level on the 34-bit FloPoCo word.
 * ============================================================================================= */

#define EXC(w)  ((uint32_t)(((w) >> 32) & 3u))
#define SGN(w)  ((uint32_t)(((w) >> 31) & 1u))
#define EXPF(w) ((uint32_t)(((w) >> 23) & 0xFFu))
#define FRAC(w) ((uint32_t)((w) & 0x7FFFFFu))

/* Wrapper used at every adder input that comes from a raw 32-bit word: exception field = {0, |bits}
 * (core/FPAddersReduceTree.sv:94-95, core/FPAggregator.v:124, ResultsCombiner.sv:295-296). */
uint64_t orc_fp34_wrap(uint32_t bits) { return ((uint64_t)(bits != 0u) << 32) | bits; }

/* Exception 00 forces the 32-bit output to +0 (FPAddersReduceTree.sv:141, FPAggregator.v:107-112). */
uint32_t orc_fp34_unwrap(uint64_t w) { return EXC(w) == 0u ? 0u : (uint32_t)w; }

uint64_t orc_fp34_add(uint64_t X, uint64_t Y) {
  /* exponent difference and swap (FPAdder...v:296-303): order by {exc, exp, frac} */
  const uint64_t keyX = ((uint64_t)EXC(X) << 31) | (X & 0x7FFFFFFFu);
  const uint64_t keyY = ((uint64_t)EXC(Y) << 31) | (Y & 0x7FFFFFFFu);
  const int swap = keyX < keyY;
  const uint64_t nX = swap ? Y : X, nY = swap ? X : Y;
  const uint32_t expX = EXPF(nX), excX = EXC(nX), excY = EXC(nY);
  const uint32_t sX = SGN(nX), sY = SGN(nY);
  const uint32_t effSub = sX ^ sY; /* :308 */

  /* exception of the result from the operands' exceptions (:313-320) */
  uint32_t excRt;
  const uint32_t ee = (excX << 2) | excY;
  if (ee == 0x0) excRt = 0;                                   /* zero + zero                 */
  else if (ee == 0x5 || ee == 0x4 || ee == 0x1) excRt = 1;    /* normal/zero mixes           */
  else if (ee == 0xA) excRt = (sX == sY) ? 2u : 3u;           /* inf+inf: same sign inf, else NaN */
  else if (ee == 0x8 || ee == 0x2 || ee == 0x9 || ee == 0x6) excRt = 2; /* inf with zero/normal */
  else excRt = 3;                                             /* anything with NaN           */
  /* sign: (+0)+(-0) in either order gives +0, otherwise sign of the larger operand (:322) */
  const uint32_t signR = (ee == 0x0 && sX != sY) ? 0u : sX;

  /* alignment (:324-330): 9-bit exponent difference of the swapped operands */
  const uint32_t expDiff = (expX - EXPF(nY)) & 0x1FFu;
  const int shiftedOut = expDiff >= 25u;
  const uint32_t shiftVal = shiftedOut ? 26u : (expDiff & 31u);
  const uint32_t fracY = (excY == 0u) ? 0u : (0x800000u | FRAC(nY)); /* :311 */
  const uint64_t shifted = (((uint64_t)fracY) << 26) >> shiftVal;    /* 50-bit RightShifter (:31-59) */

  /* far-path add with sticky (:333-344) */
  const uint32_t sticky = (shifted & 0xFFFFFFu) != 0u;
  uint32_t fracYfar = (uint32_t)((shifted >> 24) & 0x3FFFFFFu);      /* 27 bits with a leading 0 */
  if (effSub) fracYfar = (~fracYfar) & 0x7FFFFFFu;
  const uint32_t fracXfar = (1u << 25) | (FRAC(nX) << 2);            /* {01, frac, 00} */
  const uint32_t cin = effSub & (sticky ^ 1u);
  const uint32_t fracAdd = (fracXfar + fracYfar + cin) & 0x7FFFFFFu;
  uint32_t I = (fracAdd << 1) | sticky;                              /* fracGRS, 28 bits */
  const uint32_t extExpInc = expX + 1u;                              /* 10 bits */

  /* leading-zero count + normalise (LZCShifter_28_to_28_counting_32, :126-171) */
  const uint32_t M28 = 0xFFFFFFFu;
  uint32_t c4 = ((I >> 12) & 0xFFFFu) == 0u; if (c4) I = (I << 16) & M28;
  uint32_t c3 = ((I >> 20) & 0xFFu) == 0u;   if (c3) I = (I << 8) & M28;
  uint32_t c2 = ((I >> 24) & 0xFu) == 0u;    if (c2) I = (I << 4) & M28;
  uint32_t c1 = ((I >> 26) & 0x3u) == 0u;    if (c1) I = (I << 2) & M28;
  uint32_t c0 = ((I >> 27) & 0x1u) == 0u;    if (c0) I = (I << 1) & M28;
  const uint32_t nZeros = (c4 << 4) | (c3 << 3) | (c2 << 2) | (c1 << 1) | c0;
  const uint32_t updatedExp = (extExpInc - nZeros) & 0x3FFu;         /* :348 */
  const int eqdiffsign = nZeros == 31u;                              /* :349 */

  /* round to nearest even (:350-366) */
  const uint64_t expFrac = ((uint64_t)updatedExp << 24) | ((I >> 3) & 0xFFFFFFu);
  const uint32_t stk = (I & 3u) != 0u, rnd = (I >> 2) & 1u, grd = (I >> 3) & 1u, lsb = (I >> 4) & 1u;
  const uint32_t addToRound = !(lsb == 0u && grd == 1u && rnd == 0u && stk == 0u);
  const uint64_t rounded = (expFrac + addToRound) & 0x3FFFFFFFFull;
  const uint32_t upExc = (uint32_t)((rounded >> 32) & 3u);
  const uint32_t fracR = (uint32_t)((rounded >> 1) & 0x7FFFFFu);
  const uint32_t expR = (uint32_t)((rounded >> 24) & 0xFFu);

  /* exponent overflow/underflow folded into the exception field (:372-385) */
  uint32_t excRt2;
  if (excRt == 0u) excRt2 = 0;                                  /* 0000,0100,1000,1100 */
  else if (excRt == 1u) excRt2 = (upExc == 0u) ? 1u : (upExc == 1u) ? 2u : 0u; /* ok / overflow->inf / underflow->zero */
  else if (excRt == 2u) excRt2 = (upExc <= 1u) ? 2u : 3u;       /* 0010,0110 -> inf, else NaN */
  else excRt2 = 3;
  const uint32_t excR = (eqdiffsign && effSub) ? 0u : excRt2;   /* exact cancellation -> zero */
  return ((uint64_t)excR << 32) | ((uint64_t)signR << 31) | ((uint64_t)expR << 23) | fracR;
}

uint32_t orc_fpadd_bits(uint32_t a, uint32_t b) {
  return orc_fp34_unwrap(orc_fp34_add(orc_fp34_wrap(a), orc_fp34_wrap(b)));
}

void orc_fpadd_bits_batch(const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i) out[i] = orc_fpadd_bits(a[i], b[i]);
}

/* =============================================================================================
 * 2. Wire format (A2: little-endian word packing, core/PipelinedMUX.sv:65)
 * ============================================================================================= */

uint32_t orc_weights_lines_per_tree(uint32_t D) { return (uint32_t)((((1ull << (D + 1)) - 1) + 3) / 4); }
uint32_t orc_findex_lines_per_tree(uint32_t D) { return (uint32_t)((((1ull << D) - 1) + 7) / 8); }
uint32_t orc_tuple_lines(uint32_t F) { return (F + 3) / 4; }

void orc_pack_model(uint32_t T, uint32_t D, const uint32_t* thr_bits, const uint16_t* fidx,
                    const uint8_t* miss_right, const uint32_t* leaf_bits, uint32_t* wlines,
                    uint16_t* flines) {
  const uint32_t nint = (1u << D) - 1u, nleaf = 1u << D;
  const uint32_t wstride = orc_weights_lines_per_tree(D) * 4u, fstride = orc_findex_lines_per_tree(D) * 8u;
  memset(wlines, 0, (size_t)T * wstride * sizeof(uint32_t));
  memset(flines, 0, (size_t)T * fstride * sizeof(uint16_t));
  for (uint32_t i = 0; i < T; ++i) {
    uint32_t* w = wlines + (size_t)i * wstride;
    uint16_t* f = flines + (size_t)i * fstride;
    for (uint32_t n = 0; n < nint; ++n) {
      w[n] = thr_bits[(size_t)i * nint + n];
      /* entry layout DTPU.sv:628,637,659: [10:0] feature index, [13] missing-goes-right */
      f[n] = (uint16_t)((fidx[(size_t)i * nint + n] & 0x7FFu) | ((miss_right[(size_t)i * nint + n] & 1u) << 13));
    }
    for (uint32_t l = 0; l < nleaf; ++l) w[nint + l] = leaf_bits[(size_t)i * nleaf + l];
  }
}

/* =============================================================================================
 * 3. Traversal (core/DTPU.sv:579-760)
 * ============================================================================================= */

static inline int orc_is_nan_bits(uint32_t b) { return (b & 0x7FFFFFFFu) > 0x7F800000u; }

static inline int orc_less(uint32_t f, uint32_t w, uint32_t cmp_mode) {
  if (cmp_mode == 0u) {
    /* DTPU.sv:655: {~f[31],f} < {~w[31],w} as 33-bit unsigned == signed int32 compare of raw bits */
    return (int32_t)f < (int32_t)w;
  }
  /* cmp_mode 1 (this repo's extension): IEEE-754 binary32 '<' (false if either is NaN, -0 == +0) */
  if (orc_is_nan_bits(f) || orc_is_nan_bits(w)) return 0;
  float a, b;
  memcpy(&a, &f, 4);
  memcpy(&b, &w, 4);
  return a < b;
}

/* The comparison stage, DTPU.sv:653-667: incrementNodeOffset = isFeatureMissing ? isMissingRight : ~isFeatureSmaller.
 * tests/test_oracle_adder.py checks it against vectors produced by evaluating those RTL assigns themselves
 * (tests/golden/make_rtl_golden.py). */
uint32_t orc_go_right(uint32_t f, uint32_t w, uint32_t missing_bits, uint32_t miss_right, uint32_t cmp_mode) {
  if (f == missing_bits) return miss_right & 1u;      /* :653,667 bit-equality with MissingFeatureValue */
  return (uint32_t)!orc_less(f, w, cmp_mode);         /* :655-657 */
}

uint32_t orc_traverse(const orc_params* p, const uint32_t* weights_lines, const uint16_t* findex_lines,
                      const uint32_t* tuple, uint32_t tree) {
  /* stride = lines/tree (the published RTL's off-by-one stride quirk is NOT replicated) */
  const uint32_t* w = weights_lines + (size_t)tree * p->weights_lines_per_tree * 4u;
  const uint16_t* fi = findex_lines + (size_t)tree * p->findex_lines_per_tree * 8u;
  uint32_t n = 0; /* root, DTPU.sv:582 */
  for (uint32_t lvl = 0; lvl < p->num_levels; ++lvl) { /* level counter 0..LastLevelIndex, :588,663 */
    const uint16_t e = fi[n];
    const uint32_t j = e & 0x7FFu;                      /* :628 */
    const uint32_t miss_right = (e >> 13) & 1u;         /* :659 (bit 13)  */
    const uint32_t f = tuple[j], thr = w[n];
    const uint32_t right = orc_go_right(f, thr, p->missing_bits, miss_right, p->cmp_mode);
    n = 2u * n + 1u + right;                            /* :594-596,710-712 */
  }
  return w[n];                                          /* leaf read through port B, :731-732 */
}

void orc_leaves(const orc_params* p, const uint32_t* weights_lines, const uint16_t* findex_lines,
                const uint32_t* tuple, uint32_t* leaves) {
  for (uint32_t i = 0; i < p->num_trees; ++i) leaves[i] = orc_traverse(p, weights_lines, findex_lines, tuple, i);
}

/* Same walk as orc_traverse for trees [first, first+cnt), eight trees in lock-step per level so that a CPU core
 * has eight independent load chains in flight (the batch scorers below use it: it is what makes the CPU baseline
 * a fair one).  tests/test_oracle_kat.py checks it against orc_traverse tree by tree. */
static void traverse_range(const orc_params* p, const uint32_t* wl, const uint16_t* fl, const uint32_t* x,
                           uint32_t first, uint32_t cnt, uint32_t* leaves) {
  const size_t ws = (size_t)p->weights_lines_per_tree * 4u, fs = (size_t)p->findex_lines_per_tree * 8u;
  const uint32_t D = p->num_levels, miss = p->missing_bits, mode = p->cmp_mode;
  uint32_t i = 0;
  for (; i + 8 <= cnt; i += 8) {
    const uint32_t* w[8];
    const uint16_t* f[8];
    uint32_t n[8];
    for (int u = 0; u < 8; ++u) {
      w[u] = wl + (size_t)(first + i + u) * ws;
      f[u] = fl + (size_t)(first + i + u) * fs;
      n[u] = 0;
    }
    for (uint32_t lvl = 0; lvl < D; ++lvl)
      for (int u = 0; u < 8; ++u) {
        const uint16_t e = f[u][n[u]];
        const uint32_t v = x[e & 0x7FFu], thr = w[u][n[u]];
        const uint32_t right = orc_go_right(v, thr, miss, (e >> 13) & 1u, mode);
        n[u] = 2u * n[u] + 1u + right;
      }
    for (int u = 0; u < 8; ++u) leaves[i + u] = w[u][n[u]];
  }
  for (; i < cnt; ++i) leaves[i] = orc_traverse(p, wl, fl, x, first + i);
}

void orc_leaves_fast(const orc_params* p, const uint32_t* weights_lines, const uint16_t* findex_lines,
                     const uint32_t* tuple, uint32_t* leaves) {
  traverse_range(p, weights_lines, findex_lines, tuple, 0, p->num_trees, leaves);
}

/* =============================================================================================
 * 4. Reference-order reduction (one device)
 *    tree i (local stream order) -> PU i%8 (Core.sv:352-357), group g=i/8 -> cluster g%C, slot g/C
 *    (Core.sv:291-304, RLS.v:41-52).  Per cluster: for slot t: s_t = 8-way pairwise tree over PUs
 *    (FPAddersReduceTree.sv:94-141), acc <- s_t + acc (FPAggregator.v:95-131).  Then the cluster
 *    results are accumulated sequentially c = 0..C-1 by the same module (Core.sv:486-541).
 * ============================================================================================= */

static inline float f_from(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t b_from(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

/* The 8-way adder tree of one PU group, FPAddersReduceTree.sv:88-141: inputs wrapped with exc = {0, |x}, three
 * levels of FPAdder_8_23 on the 34-bit values, tree_out forced to +0 when the result's exception code is 00.
 * tests/test_oracle_adder.py checks it against vectors from the elaborated RTL (tests/golden/make_rtl_golden.py). */
uint32_t orc_tree8(const uint32_t* leaf_bits) {
  uint64_t l[8];
  for (uint32_t pu = 0; pu < 8; ++pu) l[pu] = orc_fp34_wrap(leaf_bits[pu]);
  const uint64_t a0 = orc_fp34_add(l[0], l[1]), a1 = orc_fp34_add(l[2], l[3]);
  const uint64_t a2 = orc_fp34_add(l[4], l[5]), a3 = orc_fp34_add(l[6], l[7]);
  const uint64_t b0 = orc_fp34_add(a0, a1), b1 = orc_fp34_add(a2, a3);
  return orc_fp34_unwrap(orc_fp34_add(b0, b1)); /* tree_out, :141 */
}

/* The sequential accumulator, FPAggregator.v:79-131: running 34-bit value reset to 0, per input
 * running <- FPAdder(X = {0, |x, x}, Y = running), output after the last input forced to +0 on exception 00.
 * The same module accumulates the tree slots of a cluster and then the clusters (Core.sv:486-541).
 * tests/test_oracle_adder.py checks it against the datapath of the RTL (tests/golden/make_rtl_golden.py). */
uint32_t orc_aggregate(const uint32_t* x_bits, uint32_t n) {
  uint64_t acc = 0; /* prev_aggreg_value reset, FPAggregator.v:83 */
  for (uint32_t i = 0; i < n; ++i) acc = orc_fp34_add(orc_fp34_wrap(x_bits[i]), acc); /* X = new, Y = running, :124-131 */
  return orc_fp34_unwrap(acc);
}

static uint32_t reduce_flopoco(const uint32_t* leaves, uint32_t T, uint32_t C) {
  const uint32_t groups = (T + 7u) / 8u;
  const uint32_t slots = (groups + C - 1u) / C; /* trees per PU (CSR205[43:36]); extra slots are EMPTY */
  uint32_t cluster_out[8];
  uint32_t* s = (uint32_t*)malloc((size_t)(slots ? slots : 1u) * sizeof(uint32_t));
  for (uint32_t c = 0; c < C; ++c) {
    for (uint32_t t = 0; t < slots; ++t) {
      const uint32_t g = t * C + c;
      uint32_t l[8];
      for (uint32_t pu = 0; pu < 8; ++pu) {
        const uint32_t i = g * 8u + pu;
        l[pu] = i < T ? leaves[i] : 0u; /* EMPTY slot outputs 0, DTPU.sv:760 */
      }
      s[t] = orc_tree8(l);
    }
    cluster_out[c] = orc_aggregate(s, slots);  /* slot accumulate of cluster c */
  }
  free(s);
  return orc_aggregate(cluster_out, C);        /* cluster accumulate, c = 0..C-1 */
}

static uint32_t reduce_native(const uint32_t* leaves, uint32_t T, uint32_t C) {
  const uint32_t groups = (T + 7u) / 8u;
  const uint32_t slots = (groups + C - 1u) / C;
  volatile float acc[8]; /* volatile: forbid re-association / contraction by the host compiler */
  for (uint32_t c = 0; c < C; ++c) acc[c] = 0.0f;
  for (uint32_t t = 0; t < slots; ++t) {
    for (uint32_t c = 0; c < C; ++c) {
      const uint32_t g = t * C + c;
      float l[8];
      for (uint32_t pu = 0; pu < 8; ++pu) {
        const uint32_t i = g * 8u + pu;
        l[pu] = i < T ? f_from(leaves[i]) : 0.0f;
      }
      volatile float a0 = l[0] + l[1], a1 = l[2] + l[3], a2 = l[4] + l[5], a3 = l[6] + l[7];
      volatile float b0 = a0 + a1, b1 = a2 + a3;
      volatile float s = b0 + b1;
      acc[c] = s + acc[c];
    }
  }
  volatile float tot = 0.0f;
  for (uint32_t c = 0; c < C; ++c) tot = acc[c] + tot;
  return b_from(tot);
}

uint32_t orc_reduce_device(const uint32_t* leaves, uint32_t num_trees, uint32_t C, int flopoco) {
  return flopoco ? reduce_flopoco(leaves, num_trees, C) : reduce_native(leaves, num_trees, C);
}

/* =============================================================================================
 * 5. Batch scoring
 * ============================================================================================= */

/* Threads the batch scorers use by default = CPUs this process may actually run on: the OpenMP default, capped by the
 * scheduler affinity mask and by the cgroup CPU quota (a container with `cpu.max = 1600000 100000` gets 16 CPUs' worth of
 * time however many logical CPUs the box shows; running 128 threads under that quota is slower than running 16). */
int orc_hw_threads(void) {
  int n = 1;
#ifdef _OPENMP
  n = omp_get_max_threads();
#endif
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    const int a = CPU_COUNT(&set);
    if (a > 0 && a < n) n = a;
  }
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    long long quota = -1, period = 0;
    char first[32] = {0};
    if (fscanf(f, "%31s %lld", first, &period) == 2 && strcmp(first, "max") != 0) quota = atoll(first);
    fclose(f);
    if (quota > 0 && period > 0) {
      const int q = (int)((quota + period - 1) / period);
      if (q > 0 && q < n) n = q;
    }
  }
  return n > 0 ? n : 1;
}

static int check_params(const orc_params* p, size_t n_wlines, size_t n_flines) {
  if (!p || p->num_trees == 0 || p->num_levels == 0 || p->num_levels > 16 || p->num_features == 0 ||
      p->num_features > 2048)
    return -1;
  if (p->clusters_per_tuple != 1 && p->clusters_per_tuple != 2 && p->clusters_per_tuple != 4 &&
      p->clusters_per_tuple != 8)
    return -1;
  if (p->weights_lines_per_tree < orc_weights_lines_per_tree(p->num_levels)) return -2;
  if (p->findex_lines_per_tree < orc_findex_lines_per_tree(p->num_levels)) return -2;
  if (n_wlines < (size_t)p->num_trees * p->weights_lines_per_tree) return -3;
  if (n_flines < (size_t)p->num_trees * p->findex_lines_per_tree) return -3;
  return 0;
}

static uint32_t shard_sum(const uint32_t* leaves, uint32_t n, uint32_t C, int sum_mode) {
  if (sum_mode == ORC_SUM_REF_FLOPOCO) return reduce_flopoco(leaves, n, C);
  if (sum_mode == ORC_SUM_REF_NATIVE) return reduce_native(leaves, n, C);
  double a = 0.0; /* ORC_SUM_F64_SEQ */
  for (uint32_t i = 0; i < n; ++i) a += (double)f_from(leaves[i]);
  return b_from((float)a);
}

int orc_score_shard(const orc_params* p, const void* wl, size_t n_wlines, const void* fl, size_t n_flines,
                    const void* tl, size_t n_tuples, uint32_t tree_begin, uint32_t tree_end, float* out,
                    int sum_mode, int nthreads) {
  int rc = check_params(p, n_wlines, n_flines);
  if (rc) return rc;
  if (tree_begin > tree_end || tree_end > p->num_trees) return -4;
  const uint32_t* w = (const uint32_t*)wl;
  const uint16_t* f = (const uint16_t*)fl;
  const uint32_t* t = (const uint32_t*)tl;
  const uint32_t tw = orc_tuple_lines(p->num_features) * 4u, nloc = tree_end - tree_begin;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = orc_hw_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    uint32_t* leaves = (uint32_t*)malloc(sizeof(uint32_t) * (nloc ? nloc : 1));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long long r = 0; r < (long long)n_tuples; ++r) {
      const uint32_t* x = t + (size_t)r * tw;
      traverse_range(p, w, f, x, tree_begin, nloc, leaves);
      out[r] = f_from(shard_sum(leaves, nloc, p->clusters_per_tuple, sum_mode));
    }
    free(leaves);
  }
  (void)nthreads;
  return 0;
}

int orc_score(const orc_params* p, const void* wl, size_t n_wlines, const void* fl, size_t n_flines,
              const void* tl, size_t n_tuples, float* out, double* gold, int sum_mode, int n_devices,
              int nthreads) {
  int rc = check_params(p, n_wlines, n_flines);
  if (rc) return rc;
  if (n_devices < 1 || (uint32_t)n_devices > p->num_trees) return -5;
  const uint32_t* w = (const uint32_t*)wl;
  const uint16_t* f = (const uint16_t*)fl;
  const uint32_t* t = (const uint32_t*)tl;
  const uint32_t T = p->num_trees, tw = orc_tuple_lines(p->num_features) * 4u;
  const uint32_t per_dev = (T + (uint32_t)n_devices - 1u) / (uint32_t)n_devices; /* PCIeReceiver.sv:241-264 */
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = orc_hw_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    uint32_t* leaves = (uint32_t*)malloc(sizeof(uint32_t) * T);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long long r = 0; r < (long long)n_tuples; ++r) {
      const uint32_t* x = t + (size_t)r * tw;
      traverse_range(p, w, f, x, 0, T, leaves);
      /* chain reduce host -> dev1 -> ...: each hop adds local + upstream (ResultsCombiner.sv:292-311) */
      uint32_t run = 0;
      for (int d = 0; d < n_devices; ++d) {
        const uint32_t b = (uint32_t)d * per_dev, e = (b + per_dev < T) ? b + per_dev : T;
        const uint32_t part = (b < e) ? shard_sum(leaves + b, e - b, p->clusters_per_tuple, sum_mode) : 0u;
        if (d == 0) run = part;
        else if (sum_mode == ORC_SUM_REF_FLOPOCO) run = orc_fpadd_bits(part, run);
        else { volatile float s = f_from(part) + f_from(run); run = b_from(s); }
      }
      out[r] = f_from(run);
      if (gold) {
        double a = 0.0;
        for (uint32_t i = 0; i < T; ++i) a += (double)f_from(leaves[i]);
        gold[r] = a;
      }
    }
    free(leaves);
  }
  (void)nthreads;
  return 0;
}

/* Multi-class one-vs-all (BASELINE config 5; NOT in the reference, which has no classes): class k owns the
 * trees {i : i % K == k} (interleaved) or [k*T/K, (k+1)*T/K) (class-major); each class is an independent
 * ensemble reduced in reference order (optionally tree-sharded over n_devices with chain add); label =
 * argmax over classes, lowest index on ties.  class_scores (may be NULL): [K][n]. */
int orc_classify(const orc_params* p, const void* wl, size_t n_wlines, const void* fl, size_t n_flines,
                 const void* tl, size_t n_tuples, uint32_t K, int interleaved, int sum_mode, int n_devices,
                 int32_t* labels, float* class_scores) {
  int rc = check_params(p, n_wlines, n_flines);
  if (rc) return rc;
  if (K == 0 || K > p->num_trees || (!interleaved && p->num_trees % K) || n_devices < 1) return -6;
  const uint32_t* w = (const uint32_t*)wl;
  const uint16_t* f = (const uint16_t*)fl;
  const uint32_t* t = (const uint32_t*)tl;
  const uint32_t T = p->num_trees, tw = orc_tuple_lines(p->num_features) * 4u;
#ifdef _OPENMP
#pragma omp parallel num_threads(orc_hw_threads())
#endif
  {
    uint32_t* leaves = (uint32_t*)malloc(sizeof(uint32_t) * T);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long long r = 0; r < (long long)n_tuples; ++r) {
      const uint32_t* x = t + (size_t)r * tw;
      float best = 0.f;
      int32_t arg = 0;
      for (uint32_t k = 0; k < K; ++k) {
        uint32_t nk = 0;
        for (uint32_t i = 0; i < T; ++i) {
          const uint32_t cls = interleaved ? i % K : i / (T / K);
          if (cls == k) leaves[nk++] = orc_traverse(p, w, f, x, i);
        }
        const uint32_t per_dev = (nk + (uint32_t)n_devices - 1u) / (uint32_t)n_devices;
        uint32_t run = 0;
        for (int d = 0; d < n_devices; ++d) {
          const uint32_t b = (uint32_t)d * per_dev, e = (b + per_dev < nk) ? b + per_dev : nk;
          const uint32_t part = (b < e) ? shard_sum(leaves + b, e - b, p->clusters_per_tuple, sum_mode) : 0u;
          if (d == 0) run = part;
          else if (sum_mode == ORC_SUM_REF_FLOPOCO) run = orc_fpadd_bits(part, run);
          else { volatile float s = f_from(part) + f_from(run); run = b_from(s); }
        }
        const float sc = f_from(run);
        if (class_scores) class_scores[(size_t)k * n_tuples + (size_t)r] = sc;
        if (k == 0 || sc > best || (best != best && sc == sc)) { best = sc; arg = (int32_t)k; }
      }
      labels[r] = arg;
    }
    free(leaves);
  }
  return 0;
}

/* =============================================================================================
 * 6. Deterministic synthetic inputs
 * ============================================================================================= */

#define SEED_X 0x0DD7000000000001ull
#define SEED_M 0x0DD7000000000002ull

uint64_t orc_splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static inline float unit24(uint64_t h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }

void orc_gen_tuples(uint64_t row0, size_t n, uint32_t F, int dist, uint32_t missing_bits, uint32_t* out) {
  const uint32_t tw = orc_tuple_lines(F) * 4u;
  for (size_t r = 0; r < n; ++r) {
    uint32_t* x = out + r * tw;
    for (uint32_t j = 0; j < tw; ++j) x[j] = 0u;
    for (uint32_t j = 0; j < F; ++j) {
      const uint64_t h = orc_splitmix64(SEED_X + (row0 + r) * (uint64_t)F + j);
      float v = unit24(h);
      if (dist == 1) {
        v = v * 2.0f - 1.0f;
        if (((h >> 8) & 0xFFFFu) % 20u == 0u) { x[j] = missing_bits; continue; }
      }
      x[j] = b_from(v);
    }
  }
}

void orc_gen_model(uint32_t T, uint32_t D, uint32_t F, int dist, uint32_t* wlines, uint16_t* flines) {
  const uint32_t nint = (1u << D) - 1u, ntot = (1u << (D + 1)) - 1u;
  const uint32_t wstride = orc_weights_lines_per_tree(D) * 4u, fstride = orc_findex_lines_per_tree(D) * 8u;
  memset(wlines, 0, (size_t)T * wstride * sizeof(uint32_t));
  memset(flines, 0, (size_t)T * fstride * sizeof(uint16_t));
  for (uint32_t i = 0; i < T; ++i) {
    for (uint32_t n = 0; n < ntot; ++n) {
      const uint64_t g = (uint64_t)i * (1ull << (D + 1)) + n;
      const float u = unit24(orc_splitmix64(SEED_M + 3ull * g + 1ull));
      if (n < nint) {
        const uint32_t fidx = (uint32_t)(orc_splitmix64(SEED_M + 3ull * g) % F);
        const uint32_t mr = (uint32_t)(orc_splitmix64(SEED_M + 3ull * g + 2ull) & 1ull);
        const float thr = dist == 1 ? u * 2.0f - 1.0f : u;
        wlines[(size_t)i * wstride + n] = b_from(thr);
        flines[(size_t)i * fstride + n] = (uint16_t)(fidx | (mr << 13));
      } else {
        volatile float c = u - 0.5f;
        volatile float v = c * 0.2f;
        wlines[(size_t)i * wstride + n] = b_from(v);
      }
    }
  }
}

/* =============================================================================================
 * 8. The CPU baseline form of the scorer (bench.py cpu_baseline): same results as orc_score(ORC_SUM_REF_NATIVE / F64)
 *    for cmp_mode 0, organised for a cache hierarchy instead of for readability: nodes re-packed to 8-byte
 *    {threshold, feature | miss_right << 31} records, rows processed in blocks so that a group of 8 trees (16 KB at
 *    depth 8) is walked for a whole block of rows out of L1 before the next group is touched, 8 independent walks in
 *    flight per thread, branch-free direction select, the group sums folded into the per-row cluster accumulators as
 *    they appear (no per-tree leaf buffer).  tests/test_oracle_kat.py holds it to orc_score bit for bit.
 * ============================================================================================= */
typedef struct { uint32_t thr, fi; } fast_node;

int orc_score_fast(const orc_params* p, const void* wl, size_t n_wlines, const void* fl, size_t n_flines, const void* tl,
                   size_t n_tuples, float* out, int sum_mode, int nthreads) {
  int rc = check_params(p, n_wlines, n_flines);
  if (rc) return rc;
  if (p->cmp_mode != 0 || sum_mode == ORC_SUM_REF_FLOPOCO)  /* the general scorer covers those */
    return orc_score(p, wl, n_wlines, fl, n_flines, tl, n_tuples, out, NULL, sum_mode, 1, nthreads);
  const uint32_t T = p->num_trees, D = p->num_levels, nint = (1u << D) - 1u, nleaf = 1u << D;
  const uint32_t Tp = (T + 7u) & ~7u, tw = orc_tuple_lines(p->num_features) * 4u, miss = p->missing_bits;
  const size_t ws = (size_t)p->weights_lines_per_tree * 4u, fs = (size_t)p->findex_lines_per_tree * 8u;
  fast_node* nodes = (fast_node*)calloc((size_t)Tp * nint, sizeof(fast_node)); /* EMPTY trees: leaves +0 */
  uint32_t* leaf = (uint32_t*)calloc((size_t)Tp * nleaf, sizeof(uint32_t));
  if (!nodes || !leaf) { free(nodes); free(leaf); return -7; }
  for (uint32_t i = 0; i < T; ++i) {
    const uint32_t* w = (const uint32_t*)wl + (size_t)i * ws;
    const uint16_t* f = (const uint16_t*)fl + (size_t)i * fs;
    for (uint32_t n = 0; n < nint; ++n) {
      nodes[(size_t)i * nint + n].thr = w[n];
      nodes[(size_t)i * nint + n].fi = (uint32_t)(f[n] & 0x7FFu) | ((uint32_t)((f[n] >> 13) & 1u) << 31);
    }
    memcpy(leaf + (size_t)i * nleaf, w + nint, (size_t)nleaf * 4u);
  }
  const uint32_t* t = (const uint32_t*)tl;
  const uint32_t C = p->clusters_per_tuple;
  enum { RB = 256 }; /* rows per block: 32 KB of tuples at 32 features; a PU group (8 trees, 16 KB at depth 8) is walked
                        for the whole block out of L1 / L2 before the next group is touched */
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = orc_hw_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    /* per row: the C cluster accumulators of the reference-order sum (acc <- s_g + acc on cluster g % C, FPAggregator.v:124-131),
       or one fp64 accumulator; the 8 leaves of a group are folded as soon as they are known */
    float* acc = (float*)malloc(sizeof(float) * (size_t)RB * 8u);
    double* dacc = (double*)malloc(sizeof(double) * (size_t)RB);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 2)
#endif
    for (long long b0 = 0; b0 < (long long)n_tuples; b0 += RB) {
      const uint32_t rows = (uint32_t)((long long)n_tuples - b0 < RB ? (long long)n_tuples - b0 : RB);
      for (uint32_t i = 0; i < rows * 8u; ++i) acc[i] = 0.0f;
      for (uint32_t r = 0; r < rows; ++r) dacc[r] = 0.0;
      for (uint32_t t0 = 0; t0 < Tp; t0 += 8u) {
        const fast_node* nd = nodes + (size_t)t0 * nint;
        const uint32_t* lf = leaf + (size_t)t0 * nleaf;
        const uint32_t c = (t0 / 8u) % C;
        for (uint32_t r = 0; r < rows; ++r) {
          const uint32_t* x = t + ((size_t)b0 + r) * tw;
          uint32_t n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (uint32_t lvl = 0; lvl < D; ++lvl)
            for (int u = 0; u < 8; ++u) {
              const fast_node q = nd[(size_t)u * nint + n[u]];
              const uint32_t f = x[q.fi & 0x7FFu];
              const uint32_t ge = (uint32_t)!((int32_t)f < (int32_t)q.thr);   /* DTPU.sv:655-657 */
              const uint32_t right = f == miss ? q.fi >> 31 : ge;             /* DTPU.sv:653,667 */
              n[u] = 2u * n[u] + 1u + right;
            }
          float l[8];
          for (int u = 0; u < 8; ++u) l[u] = f_from(lf[(size_t)u * nleaf + (n[u] - nint)]);
          if (sum_mode == ORC_SUM_F64_SEQ) {
            for (int u = 0; u < 8; ++u)
              if (t0 + (uint32_t)u < T) dacc[r] += (double)l[u];
          } else { /* FPAddersReduceTree.sv:94-141, then the slot accumulate of cluster c */
            volatile float a0 = l[0] + l[1], a1 = l[2] + l[3], a2 = l[4] + l[5], a3 = l[6] + l[7];
            volatile float h0 = a0 + a1, h1 = a2 + a3;
            volatile float s = h0 + h1;
            volatile float na = s + acc[(size_t)r * 8u + c];
            acc[(size_t)r * 8u + c] = na;
          }
        }
      }
      for (uint32_t r = 0; r < rows; ++r) {
        if (sum_mode == ORC_SUM_F64_SEQ) {
          out[(size_t)b0 + r] = (float)dacc[r];
        } else { /* cluster accumulate c = 0..C-1 (Core.sv:486-541) */
          volatile float tot = 0.0f;
          for (uint32_t k = 0; k < C; ++k) tot = acc[(size_t)r * 8u + k] + tot;
          out[(size_t)b0 + r] = tot;
        }
      }
    }
    free(acc);
    free(dacc);
  }
  free(nodes);
  free(leaf);
  (void)nthreads;
  return 0;
}

/* =============================================================================================
 * 8b. The cache-blocked scorer for the OTHER results of orc_score: the reference's own adder (ORC_SUM_REF_FLOPOCO) and the
 *    multi-device chain (n_devices > 1; PCIeReceiver.sv:241-264 contiguous shards, ResultsCombiner.sv:292-311 local + upstream),
 *    optionally per class of a one-vs-all model.  Same walk organisation as orc_score_fast; the reduction keeps orc_tree8 /
 *    orc_aggregate's 34-bit values but takes each add through fp34_add_fast: on two NORMAL operands whose IEEE sum is normal the
 *    FloPoCo adder is the IEEE add except for the shiftedOut case (FPAdder_2cycles_latency.v:325-326), which is decided from the
 *    operands; everything else (zero / exception operands, exponent fields 0 or 255, cancellation, sub-normal results) goes through
 *    the bit-level model orc_fp34_add.  orc_fast_add_selftest + tests/test_oracle_kat.py hold the shortcut to the model (and thereby
 *    to the RTL vectors the model is pinned with); tests hold orc_score_fast_ex to orc_score / orc_classify bit for bit.
 * ============================================================================================= */
static inline uint64_t fp34_add_fast(uint64_t X, uint64_t Y) {
  if (EXC(X) == 1u && EXC(Y) == 1u) {
    const uint32_t a = (uint32_t)X, b = (uint32_t)Y;
    if (EXPF(a) - 1u < 254u && EXPF(b) - 1u < 254u) {
      volatile float sv = f_from(a) + f_from(b);
      const uint32_t r = b_from(sv);
      if (EXPF(r) - 1u < 254u) {
        const uint32_t ma = a & 0x7FFFFFFFu, mb = b & 0x7FFFFFFFu;
        const uint32_t big = ma >= mb ? a : b, small = ma >= mb ? b : a;
        /* effective subtraction, larger operand a power of two, exponents exactly 25 apart, smaller mantissa != 0:
           the alignment shift is forced to 26 (shiftedOut), the smaller operand only leaves its sticky bit, the result is the
           larger operand; IEEE gives the float just below it */
        if (((a ^ b) >> 31) && FRAC(big) == 0u && EXPF(big) - EXPF(small) == 25u && FRAC(small) != 0u)
          return (1ull << 32) | big;
        return (1ull << 32) | r;
      }
    }
  }
  return orc_fp34_add(X, Y);
}

/* mismatches of fp34_add_fast against orc_fp34_add over n structured pseudo-random operand pairs (exponent distances 0..30
   over-represented, power-of-two and all-ones mantissas, zeros, exponent fields 0 / 1 / 254 / 255, every exception code) */
uint64_t orc_fast_add_selftest(uint64_t seed, uint64_t n) {
  uint64_t bad = 0, s = seed * 0x9E3779B97F4A7C15ull + 1u;
  for (uint64_t i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const uint64_t h = s;
    uint32_t ea = (uint32_t)(h & 0xFFu), fa = (uint32_t)((h >> 8) & 0x7FFFFFu), fb = (uint32_t)((h >> 31) & 0x7FFFFFu);
    const uint32_t kind = (uint32_t)((h >> 54) & 7u), dist = (uint32_t)((h >> 57) & 31u);
    uint32_t eb = (kind & 1u) ? (uint32_t)((h >> 20) & 0xFFu) : (ea >= dist ? ea - dist : ea + dist) & 0xFFu;
    if (kind == 2u) fa = 0u;                       /* power of two */
    if (kind == 3u) fa = 0x7FFFFFu;                /* all ones */
    if (kind == 4u) { fa = 0u; fb = (fb & 1u) ? fb : 1u; }
    if (kind == 5u) { const uint32_t t = ea; ea = eb; eb = t; fb = (fb & 3u) ? fb : 0u; }
    const uint32_t a = ((uint32_t)((h >> 62) & 1u) << 31) | (ea << 23) | fa, b = ((uint32_t)((h >> 63) & 1u) << 31) | (eb << 23) | fb;
    uint64_t X = orc_fp34_wrap(a), Y = orc_fp34_wrap(b);
    if (((h >> 40) & 0x3Fu) == 0u) X = ((uint64_t)((h >> 46) & 3u) << 32) | a;   /* any exception code now and then */
    if (((h >> 41) & 0x3Fu) == 0u) Y = ((uint64_t)((h >> 48) & 3u) << 32) | b;
    bad += fp34_add_fast(X, Y) != orc_fp34_add(X, Y);
    bad += fp34_add_fast(Y, X) != orc_fp34_add(Y, X);
  }
  return bad;
}

typedef struct { uint32_t tree[8]; uint32_t cluster; uint32_t shard_end; /* 1: last group of its shard */ uint32_t extra0, extra1; /* EMPTY slots [extra0, extra1) appended then */ } ex_group;

/* num_classes 0 = plain scores into out[n]; K >= 1 = one-vs-all: class_scores [K][n] (may be NULL) and labels[n] (may be NULL) */
int orc_score_fast_ex(const orc_params* p, const void* wl, size_t n_wlines, const void* fl, size_t n_flines, const void* tl,
                      size_t n_tuples, float* out, int sum_mode, int n_devices, uint32_t num_classes, int interleaved,
                      int32_t* labels, float* class_scores, double* gold, double* gold_abs, int nthreads) {
  int rc = check_params(p, n_wlines, n_flines);
  if (rc) return rc;
  const uint32_t T = p->num_trees, D = p->num_levels, nint = (1u << D) - 1u, nleaf = 1u << D;
  const uint32_t K = num_classes ? num_classes : 1u;
  if (p->cmp_mode != 0 || sum_mode == ORC_SUM_F64_SEQ || n_devices < 1 || K > T || (num_classes && !interleaved && T % K)) return -8;
  if (!num_classes && (uint32_t)n_devices > T) return -5;
  const uint32_t C = p->clusters_per_tuple, tw = orc_tuple_lines(p->num_features) * 4u, miss = p->missing_bits;
  const int flopoco = sum_mode == ORC_SUM_REF_FLOPOCO;
  const size_t ws = (size_t)p->weights_lines_per_tree * 4u, fs = (size_t)p->findex_lines_per_tree * 8u;
  fast_node* nodes = (fast_node*)calloc((size_t)(T + 1u) * nint, sizeof(fast_node)); /* tree T = an EMPTY slot: leaves +0 */
  uint32_t* leaf = (uint32_t*)calloc((size_t)(T + 1u) * nleaf, sizeof(uint32_t));
  /* groups in walking order: class by class, device by device, the shard's PU groups in local stream order */
  ex_group* G = (ex_group*)calloc((size_t)T / 8u + (size_t)K * (size_t)n_devices + 8u, sizeof(ex_group));
  uint32_t* ids = (uint32_t*)malloc(sizeof(uint32_t) * T);
  if (!nodes || !leaf || !G || !ids) { free(nodes); free(leaf); free(G); free(ids); return -7; }
  for (uint32_t i = 0; i < T; ++i) {
    const uint32_t* w = (const uint32_t*)wl + (size_t)i * ws;
    const uint16_t* f = (const uint16_t*)fl + (size_t)i * fs;
    for (uint32_t n = 0; n < nint; ++n) {
      nodes[(size_t)i * nint + n].thr = w[n];
      nodes[(size_t)i * nint + n].fi = (uint32_t)(f[n] & 0x7FFu) | ((uint32_t)((f[n] >> 13) & 1u) << 31);
    }
    memcpy(leaf + (size_t)i * nleaf, w + nint, (size_t)nleaf * 4u);
  }
  size_t ng = 0;
  uint32_t* dev_groups = (uint32_t*)calloc((size_t)K * (size_t)n_devices, sizeof(uint32_t)); /* groups per (class, device); 0 = empty shard */
  for (uint32_t k = 0; k < K; ++k) {
    uint32_t nk = 0;
    for (uint32_t i = 0; i < T; ++i) {
      const uint32_t cls = !num_classes ? 0u : interleaved ? i % K : i / (T / K);
      if (cls == k) ids[nk++] = i;
    }
    const uint32_t per_dev = (nk + (uint32_t)n_devices - 1u) / (uint32_t)n_devices;
    for (int d = 0; d < n_devices; ++d) {
      const uint32_t b = (uint32_t)d * per_dev < nk ? (uint32_t)d * per_dev : nk, e = (b + per_dev < nk) ? b + per_dev : nk;
      const uint32_t groups = (e - b + 7u) / 8u, slots = (groups + C - 1u) / C;
      dev_groups[(size_t)k * n_devices + d] = groups;
      for (uint32_t g = 0; g < groups; ++g) {
        ex_group* q = &G[ng++];
        for (uint32_t u = 0; u < 8u; ++u) q->tree[u] = b + g * 8u + u < e ? ids[b + g * 8u + u] : T;
        q->cluster = g % C;
        q->shard_end = g + 1u == groups;
        q->extra0 = groups;
        q->extra1 = slots * C;
      }
    }
  }
  const uint32_t* t = (const uint32_t*)tl;
  enum { RB = 256 };
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = orc_hw_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    uint64_t* acc = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)RB * 8u);  /* per row, per cluster: 34-bit running value (native: fp32 bits) */
    uint32_t* run = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)RB);       /* chain over the devices */
    float* best = (float*)malloc(sizeof(float) * (size_t)RB);
    int32_t* arg = (int32_t*)malloc(sizeof(int32_t) * (size_t)RB);
    double* gsum = (double*)malloc(sizeof(double) * (size_t)RB * 2u); /* per row: fp64 sum of the leaves in walking order, and of their magnitudes */
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 2)
#endif
    for (long long b0 = 0; b0 < (long long)n_tuples; b0 += RB) {
      const uint32_t rows = (uint32_t)((long long)n_tuples - b0 < RB ? (long long)n_tuples - b0 : RB);
      size_t gi = 0;
      for (uint32_t k = 0; k < K; ++k) {
        for (uint32_t i = 0; i < rows * 2u; ++i) gsum[i] = 0.0;
        for (int d = 0; d < n_devices; ++d) {
          const uint32_t groups = dev_groups[(size_t)k * n_devices + d];
          for (uint32_t i = 0; i < rows * 8u; ++i) acc[i] = 0u; /* prev_aggreg_value reset (FPAggregator.v:83); native: +0 */
          for (uint32_t g = 0; g < groups; ++g, ++gi) {
            const ex_group* q = &G[gi];
            const uint32_t c = q->cluster;
            for (uint32_t r = 0; r < rows; ++r) {
              const uint32_t* x = t + ((size_t)b0 + r) * tw;
              uint32_t n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
              for (uint32_t lvl = 0; lvl < D; ++lvl)
                for (int u = 0; u < 8; ++u) {
                  const fast_node nd = nodes[(size_t)q->tree[u] * nint + n[u]];
                  const uint32_t f = x[nd.fi & 0x7FFu];
                  const uint32_t ge = (uint32_t)!((int32_t)f < (int32_t)nd.thr);   /* DTPU.sv:655-657 */
                  const uint32_t right = f == miss ? nd.fi >> 31 : ge;             /* DTPU.sv:653,667 */
                  n[u] = 2u * n[u] + 1u + right;
                }
              uint32_t l[8];
              for (int u = 0; u < 8; ++u) l[u] = leaf[(size_t)q->tree[u] * nleaf + (n[u] - nint)];
              if (gold || gold_abs)
                for (int u = 0; u < 8; ++u)
                  if (q->tree[u] < T) { /* (plain scores: tree order 0..T-1, the order of orc_score's gold) */
                    gsum[2u * r] += (double)f_from(l[u]);
                    gsum[2u * r + 1u] += fabs((double)f_from(l[u]));
                  }
              if (flopoco) { /* orc_tree8 + one step of orc_aggregate, the adds through fp34_add_fast */
                uint64_t w8[8];
                for (int u = 0; u < 8; ++u) w8[u] = orc_fp34_wrap(l[u]);
                const uint64_t a0 = fp34_add_fast(w8[0], w8[1]), a1 = fp34_add_fast(w8[2], w8[3]);
                const uint64_t a2 = fp34_add_fast(w8[4], w8[5]), a3 = fp34_add_fast(w8[6], w8[7]);
                const uint64_t h0 = fp34_add_fast(a0, a1), h1 = fp34_add_fast(a2, a3);
                const uint32_t s8 = orc_fp34_unwrap(fp34_add_fast(h0, h1));
                acc[(size_t)r * 8u + c] = fp34_add_fast(orc_fp34_wrap(s8), acc[(size_t)r * 8u + c]);
              } else {
                volatile float a0 = f_from(l[0]) + f_from(l[1]), a1 = f_from(l[2]) + f_from(l[3]);
                volatile float a2 = f_from(l[4]) + f_from(l[5]), a3 = f_from(l[6]) + f_from(l[7]);
                volatile float h0 = a0 + a1, h1 = a2 + a3;
                volatile float s8 = h0 + h1;
                volatile float na = s8 + f_from((uint32_t)acc[(size_t)r * 8u + c]);
                acc[(size_t)r * 8u + c] = b_from(na);
              }
            }
          }
          /* the shard is complete (an empty one: +0): EMPTY extra slots, cluster accumulate, chain hop */
          for (uint32_t r = 0; r < rows; ++r) {
            uint32_t part = 0u;
            if (groups) {
              const ex_group* q = &G[gi - 1u];
              uint64_t* a = acc + (size_t)r * 8u;
              if (flopoco) {
                for (uint32_t g = q->extra0; g < q->extra1; ++g) a[g % C] = fp34_add_fast(orc_fp34_wrap(0u), a[g % C]);
                uint64_t tot = 0u;
                for (uint32_t c = 0; c < C; ++c) tot = fp34_add_fast(orc_fp34_wrap(orc_fp34_unwrap(a[c])), tot);
                part = orc_fp34_unwrap(tot);
              } else {
                for (uint32_t g = q->extra0; g < q->extra1; ++g) { volatile float z = 0.0f + f_from((uint32_t)a[g % C]); a[g % C] = b_from(z); }
                volatile float tot = 0.0f;
                for (uint32_t c = 0; c < C; ++c) tot = f_from((uint32_t)a[c]) + tot;
                part = b_from(tot);
              }
            }
            if (d == 0) run[r] = part;
            else if (flopoco) run[r] = orc_fp34_unwrap(fp34_add_fast(orc_fp34_wrap(part), orc_fp34_wrap(run[r])));
            else { volatile float sv = f_from(part) + f_from(run[r]); run[r] = b_from(sv); }
          }
        }
        for (uint32_t r = 0; r < rows; ++r) {
          const float sc = f_from(run[r]);
          if (gold) gold[(size_t)k * n_tuples + (size_t)b0 + r] = gsum[2u * r];
          if (gold_abs) gold_abs[(size_t)k * n_tuples + (size_t)b0 + r] = gsum[2u * r + 1u];
          if (!num_classes) { out[(size_t)b0 + r] = sc; continue; }
          if (class_scores) class_scores[(size_t)k * n_tuples + (size_t)b0 + r] = sc;
          if (k == 0 || sc > best[r] || (best[r] != best[r] && sc == sc)) { best[r] = sc; arg[r] = (int32_t)k; }
          if (k + 1u == K && labels) labels[(size_t)b0 + r] = arg[r];
        }
      }
    }
    free(acc); free(run); free(best); free(arg); free(gsum);
  }
  free(nodes); free(leaf); free(G); free(ids); free(dev_groups);
  (void)nthreads;
  return 0;
}

/* =============================================================================================
 * 7. Sparse (explicit-children) model stream -- this repository's extension, see ddt_oracle.h.
 *    Compare rule = orc_go_right (DTPU.sv:653-667); entry bits [10:0], [13] as in DTPU.sv:628,659; bit 14 keeps the
 *    RTL's meaning "the next node is a leaf" (DTPU.sv:661) for the left branch, bit 15 says it for the right one.
 * ============================================================================================= */

int orc_sparse_check(const orc_params* p, const uint32_t* lines, size_t n_lines, const uint64_t* first) {
  if (!p || !lines || !first || p->num_trees == 0 || p->num_levels == 0 || p->num_levels > 64 || p->num_features == 0 ||
      p->num_features > 2048)
    return -1;
  if (first[0] != 0 || first[p->num_trees] > n_lines) return -2;
  for (uint32_t i = 0; i < p->num_trees; ++i) {
    if (first[i + 1] <= first[i]) return -2; /* every tree has at least one line */
    const uint64_t cnt = first[i + 1] - first[i];
    const uint32_t* t = lines + first[i] * 4u;
    uint8_t* depth = (uint8_t*)calloc(cnt, 1);
    for (uint64_t n = 0; n < cnt; ++n) {
      const uint32_t e = t[4 * n + 1];
      if ((e & 0x7FFu) >= p->num_features || (e >> 16)) { free(depth); return -3; }
      if ((uint32_t)depth[n] + 1u > p->num_levels) { free(depth); return -5; }
      for (int side = 0; side < 2; ++side) {
        if ((e >> (14 + side)) & 1u) continue;
        const uint32_t c = t[4 * n + 2 + side];
        if (c <= n || c >= cnt) { free(depth); return -4; } /* children after their parent: the walk terminates */
        depth[c] = (uint8_t)(depth[n] + 1u);
      }
    }
    free(depth);
  }
  return 0;
}

uint32_t orc_traverse_sparse(const orc_params* p, const uint32_t* lines, const uint64_t* first, const uint32_t* x,
                             uint32_t tree) {
  const uint32_t* t = lines + first[tree] * 4u;
  uint32_t n = 0;
  for (;;) {
    const uint32_t* r = t + 4u * n;
    const uint32_t e = r[1];
    const uint32_t right = orc_go_right(x[e & 0x7FFu], r[0], p->missing_bits, (e >> 13) & 1u, p->cmp_mode);
    const uint32_t child = r[2u + right];
    if ((e >> (14u + right)) & 1u) return child; /* leaf value bits */
    n = child;
  }
}

/* mean number of node visits per (tuple, tree): the leaf depth actually walked (bench.py / tools: node-visit rates) */
double orc_sparse_mean_depth(const orc_params* p, const uint32_t* lines, const uint64_t* first, const uint32_t* tuples, size_t n) {
  const uint32_t tw = orc_tuple_lines(p->num_features) * 4u;
  double visits = 0.0;
  for (size_t r = 0; r < n; ++r) {
    const uint32_t* x = tuples + r * tw;
    for (uint32_t i = 0; i < p->num_trees; ++i) {
      const uint32_t* t = lines + first[i] * 4u;
      uint32_t node = 0;
      for (;;) {
        const uint32_t* q = t + 4u * node;
        const uint32_t right = orc_go_right(x[q[1] & 0x7FFu], q[0], p->missing_bits, (q[1] >> 13) & 1u, p->cmp_mode);
        visits += 1.0;
        if ((q[1] >> (14u + right)) & 1u) break;
        node = q[2u + right];
      }
    }
  }
  return n ? visits / ((double)n * p->num_trees) : 0.0;
}

int orc_score_sparse(const orc_params* p, const void* nl, size_t n_lines, const uint64_t* first, const void* tl,
                     size_t n_tuples, float* out, double* gold, int sum_mode, int n_devices, int nthreads) {
  int rc = orc_sparse_check(p, (const uint32_t*)nl, n_lines, first);
  if (rc) return rc;
  const uint32_t C = p->clusters_per_tuple;
  if (C != 1 && C != 2 && C != 4 && C != 8) return -1;
  if (n_devices < 1 || (uint32_t)n_devices > p->num_trees) return -5;
  const uint32_t* lines = (const uint32_t*)nl;
  const uint32_t* t = (const uint32_t*)tl;
  const uint32_t T = p->num_trees, tw = orc_tuple_lines(p->num_features) * 4u;
  const uint32_t per_dev = (T + (uint32_t)n_devices - 1u) / (uint32_t)n_devices;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = orc_hw_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    uint32_t* leaves = (uint32_t*)malloc(sizeof(uint32_t) * T);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long long r = 0; r < (long long)n_tuples; ++r) {
      const uint32_t* x = t + (size_t)r * tw;
      for (uint32_t i = 0; i < T; ++i) leaves[i] = orc_traverse_sparse(p, lines, first, x, i);
      uint32_t run = 0;
      for (int d = 0; d < n_devices; ++d) { /* same device model as orc_score */
        const uint32_t b = (uint32_t)d * per_dev, e = (b + per_dev < T) ? b + per_dev : T;
        const uint32_t part = (b < e) ? shard_sum(leaves + b, e - b, C, sum_mode) : 0u;
        if (d == 0) run = part;
        else if (sum_mode == ORC_SUM_REF_FLOPOCO) run = orc_fpadd_bits(part, run);
        else { volatile float s = f_from(part) + f_from(run); run = b_from(s); }
      }
      out[r] = f_from(run);
      if (gold) {
        double a = 0.0;
        for (uint32_t i = 0; i < T; ++i) a += (double)f_from(leaves[i]);
        gold[r] = a;
      }
    }
    free(leaves);
  }
  (void)nthreads;
  return 0;
}

/* The CPU-baseline form of orc_score_sparse (bench.py --config 4 cpu_baseline; single device, ORC_SUM_REF_NATIVE or
 * ORC_SUM_F64_SEQ): identical results, organised for caches.  A deep forest is ~100 MB, so walking all trees for one
 * row at a time misses the cache on nearly every visit; here a block of rows (their tuples stay in L2) is taken through
 * ONE tree at a time (the tree, ~200 KB, stays in L2), eight rows in flight per thread, and the eight leaves of a PU
 * group are folded into the per-row cluster accumulators as soon as the group is complete.
 * tests/test_sparse.py holds it to orc_score_sparse bit for bit. */
int orc_score_sparse_fast(const orc_params* p, const void* nl, size_t n_lines, const uint64_t* first, const void* tl,
                          size_t n_tuples, float* out, int sum_mode, int nthreads) {
  if (sum_mode == ORC_SUM_REF_FLOPOCO) return orc_score_sparse(p, nl, n_lines, first, tl, n_tuples, out, NULL, sum_mode, 1, nthreads);
  int rc = orc_sparse_check(p, (const uint32_t*)nl, n_lines, first);
  if (rc) return rc;
  const uint32_t C = p->clusters_per_tuple;
  if (C != 1 && C != 2 && C != 4 && C != 8) return -1;
  const uint32_t* lines = (const uint32_t*)nl;
  const uint32_t* t = (const uint32_t*)tl;
  const uint32_t T = p->num_trees, tw = orc_tuple_lines(p->num_features) * 4u, miss = p->missing_bits, mode = p->cmp_mode;
  enum { RB = 1024 };
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = orc_hw_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    float* acc = (float*)malloc(sizeof(float) * (size_t)RB * 8u);
    double* dacc = (double*)malloc(sizeof(double) * (size_t)RB);
    float* grp = (float*)malloc(sizeof(float) * (size_t)RB * 8u); /* leaves of the current PU group, [row][tree in group] */
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (long long b0 = 0; b0 < (long long)n_tuples; b0 += RB) {
      const uint32_t rows = (uint32_t)((long long)n_tuples - b0 < RB ? (long long)n_tuples - b0 : RB);
      for (uint32_t i = 0; i < rows * 8u; ++i) acc[i] = 0.0f;
      for (uint32_t r = 0; r < rows; ++r) dacc[r] = 0.0;
      for (uint32_t t0 = 0; t0 < T; t0 += 8u) {
        const uint32_t c = (t0 / 8u) % C;
        for (uint32_t u = 0; u < 8u; ++u) {
          if (t0 + u >= T) { /* EMPTY slot: +0 (DTPU.sv:544,760) */
            for (uint32_t r = 0; r < rows; ++r) grp[(size_t)r * 8u + u] = 0.0f;
            continue;
          }
          const uint32_t* tr = lines + first[t0 + u] * 4u;
          uint32_t r = 0;
          for (; r + 8u <= rows; r += 8u) { /* eight rows in flight through this tree */
            uint32_t node[8], done = 0;
            for (int k = 0; k < 8; ++k) node[k] = 0;
            while (done != 0xFFu)
              for (int k = 0; k < 8; ++k) {
                if ((done >> k) & 1u) continue;
                const uint32_t* q = tr + 4u * node[k];
                const uint32_t* x = t + ((size_t)b0 + r + (uint32_t)k) * tw;
                const uint32_t right = orc_go_right(x[q[1] & 0x7FFu], q[0], miss, (q[1] >> 13) & 1u, mode);
                const uint32_t child = q[2u + right];
                if ((q[1] >> (14u + right)) & 1u) {
                  grp[((size_t)r + (uint32_t)k) * 8u + u] = f_from(child);
                  done |= 1u << k;
                } else {
                  node[k] = child;
                }
              }
          }
          for (; r < rows; ++r) {
            const uint32_t* x = t + ((size_t)b0 + r) * tw;
            uint32_t node = 0;
            for (;;) {
              const uint32_t* q = tr + 4u * node;
              const uint32_t right = orc_go_right(x[q[1] & 0x7FFu], q[0], miss, (q[1] >> 13) & 1u, mode);
              if ((q[1] >> (14u + right)) & 1u) { grp[(size_t)r * 8u + u] = f_from(q[2u + right]); break; }
              node = q[2u + right];
            }
          }
        }
        for (uint32_t r = 0; r < rows; ++r) {
          const float* l = grp + (size_t)r * 8u;
          if (sum_mode == ORC_SUM_F64_SEQ) {
            for (uint32_t u = 0; u < 8u && t0 + u < T; ++u) dacc[r] += (double)l[u];
          } else {
            volatile float a0 = l[0] + l[1], a1 = l[2] + l[3], a2 = l[4] + l[5], a3 = l[6] + l[7];
            volatile float h0 = a0 + a1, h1 = a2 + a3;
            volatile float s = h0 + h1;
            volatile float na = s + acc[(size_t)r * 8u + c];
            acc[(size_t)r * 8u + c] = na;
          }
        }
      }
      for (uint32_t r = 0; r < rows; ++r) {
        if (sum_mode == ORC_SUM_F64_SEQ) {
          out[(size_t)b0 + r] = (float)dacc[r];
        } else {
          volatile float tot = 0.0f;
          for (uint32_t k = 0; k < C; ++k) tot = acc[(size_t)r * 8u + k] + tot;
          out[(size_t)b0 + r] = tot;
        }
      }
    }
    free(acc);
    free(dacc);
    free(grp);
  }
  (void)nthreads;
  return 0;
}

/* one-vs-all classes over a sparse stream: the sparse counterpart of orc_classify (same class rule, same per-class
 * device model, same argmax: lowest index wins ties, a NaN never beats a number) */
int orc_classify_sparse(const orc_params* p, const void* nl, size_t n_lines, const uint64_t* first, const void* tl, size_t n_tuples,
                        uint32_t K, int interleaved, int sum_mode, int n_devices, int32_t* labels, float* class_scores) {
  int rc = orc_sparse_check(p, (const uint32_t*)nl, n_lines, first);
  if (rc) return rc;
  const uint32_t C = p->clusters_per_tuple;
  if (C != 1 && C != 2 && C != 4 && C != 8) return -1;
  if (K == 0 || K > p->num_trees || (!interleaved && p->num_trees % K) || n_devices < 1) return -6;
  const uint32_t* lines = (const uint32_t*)nl;
  const uint32_t* t = (const uint32_t*)tl;
  const uint32_t T = p->num_trees, tw = orc_tuple_lines(p->num_features) * 4u;
#ifdef _OPENMP
#pragma omp parallel num_threads(orc_hw_threads())
#endif
  {
    uint32_t* leaves = (uint32_t*)malloc(sizeof(uint32_t) * T);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long long r = 0; r < (long long)n_tuples; ++r) {
      const uint32_t* x = t + (size_t)r * tw;
      float best = 0.f;
      int32_t arg = 0;
      for (uint32_t k = 0; k < K; ++k) {
        uint32_t nk = 0;
        for (uint32_t i = 0; i < T; ++i) {
          const uint32_t cls = interleaved ? i % K : i / (T / K);
          if (cls == k) leaves[nk++] = orc_traverse_sparse(p, lines, first, x, i);
        }
        const uint32_t per_dev = (nk + (uint32_t)n_devices - 1u) / (uint32_t)n_devices;
        uint32_t run = 0;
        for (int d = 0; d < n_devices; ++d) {
          const uint32_t b = (uint32_t)d * per_dev, e = (b + per_dev < nk) ? b + per_dev : nk;
          const uint32_t part = (b < e) ? shard_sum(leaves + b, e - b, C, sum_mode) : 0u;
          if (d == 0) run = part;
          else if (sum_mode == ORC_SUM_REF_FLOPOCO) run = orc_fpadd_bits(part, run);
          else { volatile float s = f_from(part) + f_from(run); run = b_from(s); }
        }
        const float sc = f_from(run);
        if (class_scores) class_scores[(size_t)k * n_tuples + (size_t)r] = sc;
        if (k == 0 || sc > best || (best != best && sc == sc)) { best = sc; arg = (int32_t)k; }
      }
      labels[r] = arg;
    }
    free(leaves);
  }
  return 0;
}

void orc_sparse_from_perfect(const orc_params* p, const uint32_t* wl, const uint16_t* fl, uint32_t* lines, uint64_t* first) {
  const uint32_t D = p->num_levels, nint = (1u << D) - 1u;
  const size_t ws = (size_t)p->weights_lines_per_tree * 4u, fs = (size_t)p->findex_lines_per_tree * 8u;
  for (uint32_t i = 0; i < p->num_trees; ++i) {
    first[i] = (uint64_t)i * nint;
    const uint32_t* w = wl + (size_t)i * ws;
    const uint16_t* f = fl + (size_t)i * fs;
    uint32_t* t = lines + (size_t)i * nint * 4u;
    for (uint32_t n = 0; n < nint; ++n) {
      const int last = 2u * n + 1u >= nint; /* children are leaves */
      t[4 * n + 0] = w[n];
      t[4 * n + 1] = (uint32_t)(f[n] & 0x27FFu) | (last ? 0xC000u : 0u);
      t[4 * n + 2] = last ? w[2u * n + 1u] : 2u * n + 1u;
      t[4 * n + 3] = last ? w[2u * n + 2u] : 2u * n + 2u;
    }
  }
  first[p->num_trees] = (uint64_t)p->num_trees * nint;
}

static int sparse_fill(const uint32_t* t, uint32_t D, uint32_t n, int is_leaf, uint32_t leaf_bits, uint32_t h,
                       uint32_t lvl, uint32_t* w, uint16_t* f) {
  const uint32_t nint = (1u << D) - 1u;
  if (lvl == D) {
    if (!is_leaf) return -1; /* deeper than D */
    w[h] = leaf_bits;
    return 0;
  }
  if (is_leaf) { /* dummy node: threshold 0, feature 0; both sub-trees repeat the value */
    w[h] = 0u;
    f[h] = 0u;
    int rc = sparse_fill(t, D, 0, 1, leaf_bits, 2u * h + 1u, lvl + 1u, w, f);
    return rc ? rc : sparse_fill(t, D, 0, 1, leaf_bits, 2u * h + 2u, lvl + 1u, w, f);
  }
  (void)nint;
  const uint32_t* r = t + 4u * n;
  w[h] = r[0];
  f[h] = (uint16_t)(r[1] & 0x27FFu);
  int rc = sparse_fill(t, D, r[2], (r[1] >> 14) & 1u, r[2], 2u * h + 1u, lvl + 1u, w, f);
  return rc ? rc : sparse_fill(t, D, r[3], (r[1] >> 15) & 1u, r[3], 2u * h + 2u, lvl + 1u, w, f);
}

int orc_sparse_to_perfect(const orc_params* p, const uint32_t* lines, const uint64_t* first, uint32_t* wl, uint16_t* fl) {
  const uint32_t D = p->num_levels;
  if (D == 0 || D > 16) return -1;
  const size_t ws = (size_t)orc_weights_lines_per_tree(D) * 4u, fs = (size_t)orc_findex_lines_per_tree(D) * 8u;
  memset(wl, 0, (size_t)p->num_trees * ws * 4u);
  memset(fl, 0, (size_t)p->num_trees * fs * 2u);
  for (uint32_t i = 0; i < p->num_trees; ++i) {
    int rc = sparse_fill(lines + first[i] * 4u, D, 0, 0, 0u, 0u, 0u, wl + (size_t)i * ws, fl + (size_t)i * fs);
    if (rc) return rc;
  }
  return 0;
}

#define SEED_S 0x0DD7000000000003ull

size_t orc_gen_sparse_model(uint32_t T, uint32_t max_depth, uint32_t F, uint32_t full_levels, uint32_t split_permille,
                            int dist, uint32_t* lines, size_t cap, uint64_t* first) {
  /* BFS growth; the n-th internal node of tree i in BFS order has hash base g = i << 24 | n */
  size_t total = 0;
  const uint32_t qcap = 4u * 1024u * 1024u; /* internal nodes per tree (hash base keeps 24 bits for n) */
  uint32_t* depth_q = (uint32_t*)malloc(sizeof(uint32_t) * qcap);
  for (uint32_t i = 0; i < T; ++i) {
    if (first) first[i] = total;
    uint32_t cnt = 1; /* internal nodes allocated so far (node 0 = root, depth 0) */
    depth_q[0] = 0;
    for (uint32_t n = 0; n < cnt; ++n) {
      const uint64_t g = ((uint64_t)i << 24) | n;
      const uint32_t d = depth_q[n];
      uint32_t e = (uint32_t)(orc_splitmix64(SEED_S + 8ull * g) % F) | ((uint32_t)(orc_splitmix64(SEED_S + 8ull * g + 2ull) & 1ull) << 13);
      const float u = unit24(orc_splitmix64(SEED_S + 8ull * g + 1ull));
      uint32_t child[2];
      for (uint32_t side = 0; side < 2; ++side) {
        const uint64_t hs = orc_splitmix64(SEED_S + 8ull * g + 4ull + side);
        const int internal = d + 1u < max_depth && cnt < qcap && (d + 1u < full_levels || (uint32_t)((hs >> 20) % 1000ull) < split_permille);
        if (internal) {
          depth_q[cnt] = d + 1u;
          child[side] = cnt++;
        } else {
          volatile float c = unit24(orc_splitmix64(SEED_S + 8ull * g + 6ull + side)) - 0.5f;
          volatile float v = c * 0.2f;
          child[side] = b_from(v);
          e |= 1u << (14 + side);
        }
      }
      if (lines && total + n < cap) {
        uint32_t* r = lines + (total + n) * 4u;
        r[0] = b_from(dist == 1 ? u * 2.0f - 1.0f : u);
        r[1] = e;
        r[2] = child[0];
        r[3] = child[1];
      }
    }
    total += cnt;
  }
  if (first) first[T] = total;
  free(depth_q);
  return total;
}
