/*
 * ddt_oracle.h -- CPU ORACLE for the tree-ensemble scoring hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it, and only as the checker /
 * the reported CPU baseline.  The product (libddt.so) never links, loads or calls this code.
 *
 * What it is: a plain-C restatement of the scoring semantics of the FPGA reference
 * (fpgasystems/Distributed-DecisionTrees, RTL under rtl/DTEngine/).  Every function cites the
 * reference file:line it follows.  The reference is hardware only (no host software, no CPU
 * scorer, no tests, no golden vectors) and cannot be simulated in this environment (no Verilog
 * simulator, vendor IP missing).  What could be pinned against the reference ITSELF is pinned:
 *
 *   PINNED    against golden vectors produced by EVALUATING THE REFERENCE'S OWN RTL SOURCE TEXT with the
 *             Verilog-subset evaluator of tests/golden/make_rtl_golden.py (tests/test_oracle_adder.py):
 *               orc_fp34_add   common/FPAdder_2cycles_latency.v, whole module                      17,884 vectors
 *               orc_go_right   comparison-stage assigns, core/DTPU.sv:653-667                       21,072 vectors
 *               orc_tree8      elaborated generate loops of core/FPAddersReduceTree.sv:88-141        3,000 vectors
 *               orc_aggregate  datapath of core/FPAggregator.v (wrap, adder wiring, next state, output)  700 sequences
 *               chain hop      elaborated adders of ResultsCombiner.sv:292-311 (where exception != 00; on 00 the
 *                              RTL forwards a non-zero garbage pattern -- a defect, NOT replicated: +0 here)
 *               orc_traverse   datapath of core/DTPU.sv: all assigns, the wiring of its delay pipelines, the
 *                              clocked update of the recirculating instruction; memories as flat arrays   960 walks
 *               orc_reduce_device  the tree -> PU / cluster / slot schedule and the order clusters are accumulated in:
 *                              Core.sv:167-245,291-372,503-541 and core/RLS.v:36-62 EXECUTED cycle by cycle by the
 *                              procedural-Verilog interpreter of tests/golden/make_schedule_golden.py (one documented
 *                              repair: the unreachable IDLE -> PROG_MODE edge), 16 (C, T) cases; tests/test_oracle_schedule.py
 *                              rebuilds the summation from those placements and holds orc_reduce_device to it
 *
 *               orc_leaves / orc_traverse's reading of the WIRE FORMAT (word n of a tree's weights lines = node n, leaves
 *                              behind the 2^D-1 internal nodes, u16 entry n, feature j of a tuple, stride = lines per tree,
 *                              little-endian lines, EMPTY slots = +0): ONE DTPU programmed and driven line by line --
 *                              core/DTPU.sv:304-354,379-399,429-447,459-460,512-567 executed by the interpreter, memory
 *                              wrappers core/Mem1in2out.v / dualport_mem.v, core/PipelinedMUX.sv elaborated from its
 *                              generate blocks, control word from Core.sv:380 -- tests/golden/make_program_golden.py,
 *                              222 walks over 10 PU programs; tests/test_oracle_program.py (also holds the PRODUCT's CSR
 *                              codec to the executed EngineCSR.sv:146-308)
 *
 *               orc_score(n_devices) / orc_score_shard   the per-device tree split (PCIeReceiver.sv:136-316 executed on the
 *                              registers of the executed EngineCSR.sv) and the result chain host -> dev1 -> ... (the forwarding
 *                              blocks of ResultsCombiner.sv:355-393,422-453 executed around the hop adders):
 *                              tests/golden/make_receiver_golden.py, tests/test_oracle_receiver.py
 *               orc_aggregate  also WITH the module's control: core/FPAggregator.v executed cycle by cycle (show-ahead FIFO for
 *                              the absent quick_fifo): strictly sequential from 3 cycles between inputs (closer: the published
 *                              module loses addends, a defect) -- make_program_golden.py part 5
 *
 *      *** PARITY UNPINNED for everything else *** -- valid / ready handshakes, FIFO depths and back-pressure, the
 *      time-stamp alignment of the PUs, the PCIe stream splitter, the multi-device plumbing:
 *      sequential SystemVerilog that nothing here can execute as a whole.  The SPARSE format (section further
 *      down) is this repository's extension: it has no RTL to be pinned to and is anchored to the perfect-tree
 *      oracle through pad_to_perfect instead.
 *
 * Mitigations for the unpinned part (tests/test_oracle_*.py): hand-computed known-answer tests for every
 * rule, an independent numpy restatement, and a cross-check of the traversal against scikit-learn.
 */
#ifndef DDT_ORACLE_H
#define DDT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Run parameters == the reference's CSR 204/205 fields (rtl/DTEngine/EngineCSR.sv:218-233). */
typedef struct orc_params {
  uint32_t num_trees;              /* real trees in the stream (slots beyond are EMPTY, DTPU.sv:544,760) */
  uint32_t num_levels;             /* D = compare levels per tree, CSR205[35:32]                     */
  uint32_t num_features;           /* F; tuple lines = ceil(F/4), CSR204[63:48]                      */
  uint32_t missing_bits;           /* CSR205[31:0] (DTPU.sv:445,653)                                 */
  uint32_t weights_lines_per_tree; /* CSR204[31:16]; >= ceil((2^(D+1)-1)/4)                          */
  uint32_t findex_lines_per_tree;  /* CSR204[47:32]; >= ceil((2^D-1)/8)                              */
  uint32_t cmp_mode;               /* 0 = reference int32-bit-pattern compare (DTPU.sv:655), 1 = IEEE '<' */
  uint32_t clusters_per_tuple;     /* C in {1,2,4,8}, CSR205[47:44] (Core.sv:305-316,486-541)        */
} orc_params;

/* Summation modes for orc_score(). */
enum {
  ORC_SUM_REF_FLOPOCO = 0, /* reference order, bit-level FloPoCo adder model (normative)            */
  ORC_SUM_REF_NATIVE  = 1, /* reference order, host IEEE fp32 adds (== mode 0 on normal values)     */
  ORC_SUM_F64_SEQ     = 2, /* fp64 sequential over trees in stream order, rounded once to fp32      */
};

/* ---- fp32 adder of the reference ------------------------------------------------------------ */
/* 34-bit FloPoCo word: [33:32] exception (00 zero, 01 normal, 10 inf, 11 NaN), [31] sign,
 * [30:23] exponent, [22:0] fraction.  common/FPAdder_2cycles_latency.v:210-387. */
uint64_t orc_fp34_wrap(uint32_t bits);            /* {1'b0, |bits, bits}: FPAddersReduceTree.sv:94-95 */
uint32_t orc_fp34_unwrap(uint64_t w);             /* exc==00 ? 0 : w[31:0]: FPAddersReduceTree.sv:141 */
uint64_t orc_fp34_add(uint64_t x, uint64_t y);    /* FPAdder_8_23_uid2_l2                            */
uint32_t orc_fpadd_bits(uint32_t a, uint32_t b);  /* unwrap(add(wrap(a), wrap(b)))                   */
uint32_t orc_tree8(const uint32_t* leaf_bits);        /* 8-way adder tree, FPAddersReduceTree.sv:88-141 */
uint32_t orc_aggregate(const uint32_t* x_bits, uint32_t n); /* sequential accumulator, FPAggregator.v:79-131 */
uint32_t orc_go_right(uint32_t f_bits, uint32_t w_bits, uint32_t missing_bits, uint32_t miss_right,
                      uint32_t cmp_mode);         /* comparison stage, DTPU.sv:653-667                */
void orc_fpadd_bits_batch(const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n);

/* ---- traversal ------------------------------------------------------------------------------- */
/* One (tuple, tree) walk.  tuple = F_pad fp32 words (F padded to a multiple of 4, A2 packing).
 * Returns the raw bits of the selected leaf.  DTPU.sv:579-760. */
uint32_t orc_traverse(const orc_params* p, const uint32_t* weights_lines, const uint16_t* findex_lines,
                      const uint32_t* tuple, uint32_t tree);

/* Leaf bits of every tree for one tuple (leaves[num_trees]). */
void orc_leaves(const orc_params* p, const uint32_t* weights_lines, const uint16_t* findex_lines,
                const uint32_t* tuple, uint32_t* leaves);

/* same result as orc_leaves, eight trees walked in lock-step (the batch scorers' inner loop) */
void orc_leaves_fast(const orc_params* p, const uint32_t* weights_lines, const uint16_t* findex_lines,
                     const uint32_t* tuple, uint32_t* leaves);

/* Reference-order reduction of per-tree leaf bits for ONE device (DTPUCluster.sv:188-201,
 * FPAddersReduceTree.sv:94-141, FPAggregator.v:79-131, Core.sv:486-541).  flopoco!=0 uses the
 * bit-level adder model, else host IEEE adds. */
uint32_t orc_reduce_device(const uint32_t* leaves, uint32_t num_trees, uint32_t clusters_per_tuple,
                           int flopoco);

/* Score n tuples.  tuple_lines: n * ceil(F/4) lines of 16 B (row-major fp32, zero padded).
 * n_devices >= 1 models the tree-sharded multi-FPGA mode: trees are split into contiguous shards of
 * ceil(T/n_devices) trees (PCIeReceiver.sv:241-264), each device reduces its shard in reference order,
 * and the partials are chain-added host -> dev1 -> ... (ResultsCombiner.sv:292-311,359-369).
 * out[n] receives fp32 scores; gold[n] (optional, may be NULL) receives the exact-ish fp64 sum.
 * nthreads <= 0 -> all hardware threads (OpenMP).  Returns 0, or <0 on bad parameters. */
int orc_score(const orc_params* p, const void* weights_lines, size_t n_wlines,
              const void* findex_lines, size_t n_flines, const void* tuple_lines, size_t n_tuples,
              float* out, double* gold, int sum_mode, int n_devices, int nthreads);

/* The CPU-baseline form of orc_score (single device, ORC_SUM_REF_NATIVE or ORC_SUM_F64_SEQ, cmp_mode 0): identical
 * results, cache-blocked and branch-free (see ddt_oracle.c section 8); other modes fall through to orc_score. */
int orc_score_fast(const orc_params* p, const void* weights_lines, size_t n_wlines, const void* findex_lines,
                   size_t n_flines, const void* tuple_lines, size_t n_tuples, float* out, int sum_mode, int nthreads);

/* The cache-blocked form of the rest of orc_score / orc_classify (ddt_oracle.c section 8b; cmp_mode 0, ORC_SUM_REF_NATIVE or
 * ORC_SUM_REF_FLOPOCO): any n_devices (the chain of ResultsCombiner.sv:292-311 over contiguous shards), num_classes 0 = plain scores
 * into out[n], K >= 1 = one-vs-all (labels[n], class_scores[K][n], either may be NULL); gold / gold_abs (may be NULL; [n] or [K][n]):
 * the fp64 sum of a row's leaves (plain scores: orc_score's gold, same order) and of their magnitudes.  Identical bits; used where bench.py checks
 * millions of rows of a multi-rank job or of sum_mode 2 in seconds.  orc_fast_add_selftest: mismatches of its adder shortcut against
 * orc_fp34_add over n structured pseudo-random operand pairs (must be 0). */
int orc_score_fast_ex(const orc_params* p, const void* weights_lines, size_t n_wlines, const void* findex_lines, size_t n_flines,
                      const void* tuple_lines, size_t n_tuples, float* out, int sum_mode, int n_devices, uint32_t num_classes,
                      int interleaved, int32_t* labels, float* class_scores, double* gold, double* gold_abs, int nthreads);
uint64_t orc_fast_add_selftest(uint64_t seed, uint64_t n);

/* Partial (per-shard) scores: trees [tree_begin, tree_end) only, reduced as one device. */
int orc_score_shard(const orc_params* p, const void* weights_lines, size_t n_wlines,
                    const void* findex_lines, size_t n_flines, const void* tuple_lines, size_t n_tuples,
                    uint32_t tree_begin, uint32_t tree_end, float* out, int sum_mode, int nthreads);

/* Multi-class one-vs-all with argmax (BASELINE config 5; an extension, the reference has no classes):
 * see ddt_oracle.c.  labels[n]; class_scores[K][n] may be NULL. */
int orc_classify(const orc_params* p, const void* weights_lines, size_t n_wlines, const void* findex_lines,
                 size_t n_flines, const void* tuple_lines, size_t n_tuples, uint32_t num_classes, int interleaved,
                 int sum_mode, int n_devices, int32_t* labels, float* class_scores);

/* ---- sparse (explicit-children) model stream ----------------------------------------------------
 * An EXTENSION of this repository for trees that do not fit the reference's perfect-heap format (BASELINE
 * config 4: depth-16 random forests need 2^17 words per tree when padded).  The reference's own hook for such
 * trees is its disabled hybrid path: entry bit 14 "next node is a leaf" (core/DTPU.sv:637,661,675,712-715) and
 * the PartialTrees control bit (Core.sv:380 bit 8, DTPU.sv:20-28,736-745) -- ill-defined in the published RTL
 * (SURVEY A10b), so the format below is defined HERE and its oracle is pinned to the perfect-tree oracle through
 * orc_sparse_to_perfect (pad) + orc_traverse (tests/test_sparse.py): same compare rule (orc_go_right), same
 * reduction.  One 128-bit line per INTERNAL node (A2 packing, PipelinedMUX.sv:65):
 *   word 0        threshold bits
 *   word 1[15:0]  feature-index entry, reference bit layout (DTPU.sv:628,637,659-661): [10:0] feature,
 *                 [13] missing goes right, [14] the LEFT child is a leaf, [15] the RIGHT child is a leaf
 *   word 2, 3     left / right child: node index relative to the tree's first line, or the leaf's fp32 bits
 * Children have larger indices than their parent (BFS, DFS pre-order, ...).  tree_first_line[T+1] delimits the
 * trees; a tree that is a single leaf is one line with both leaf flags set and the value twice.
 * orc_params.num_levels = upper bound of the depth (1..64); the lines-per-tree fields are ignored. */
int orc_sparse_check(const orc_params* p, const uint32_t* node_lines, size_t n_lines, const uint64_t* tree_first_line);
uint32_t orc_traverse_sparse(const orc_params* p, const uint32_t* node_lines, const uint64_t* tree_first_line,
                             const uint32_t* tuple, uint32_t tree);
double orc_sparse_mean_depth(const orc_params* p, const uint32_t* node_lines, const uint64_t* tree_first_line,
                             const uint32_t* tuples, size_t n_tuples);
/* same reduction / multi-device model as orc_score */
int orc_score_sparse(const orc_params* p, const void* node_lines, size_t n_lines, const uint64_t* tree_first_line,
                     const void* tuple_lines, size_t n_tuples, float* out, double* gold, int sum_mode, int n_devices,
                     int nthreads);
/* one-vs-all classes over a sparse stream (see orc_classify); labels[n], class_scores[K][n] may be NULL */
int orc_classify_sparse(const orc_params* p, const void* node_lines, size_t n_lines, const uint64_t* tree_first_line,
                        const void* tuple_lines, size_t n_tuples, uint32_t num_classes, int interleaved, int sum_mode,
                        int n_devices, int32_t* labels, float* class_scores);
/* cache-blocked form of orc_score_sparse for the CPU baseline (single device; see ddt_oracle.c) */
int orc_score_sparse_fast(const orc_params* p, const void* node_lines, size_t n_lines, const uint64_t* tree_first_line,
                          const void* tuple_lines, size_t n_tuples, float* out, int sum_mode, int nthreads);
/* perfect stream -> sparse stream (one line per internal node, heap order); node_lines: T*(2^D-1) lines */
void orc_sparse_from_perfect(const orc_params* p, const uint32_t* wlines, const uint16_t* flines, uint32_t* node_lines,
                             uint64_t* tree_first_line);
/* pad_to_perfect: a leaf above depth D = p->num_levels becomes a dummy sub-tree repeating its value.
 * wlines / flines sized by orc_*_lines_per_tree(D); returns <0 if a tree is deeper than D */
int orc_sparse_to_perfect(const orc_params* p, const uint32_t* node_lines, const uint64_t* tree_first_line,
                          uint32_t* wlines, uint16_t* flines);
/* synthetic random-forest-like sparse model (deterministic): every node above `full_levels` is internal, below it a
 * child is internal with probability split_permille/1000 until max_depth.  Returns the number of lines (also when
 * node_lines == NULL or cap is too small: call twice). */
size_t orc_gen_sparse_model(uint32_t T, uint32_t max_depth, uint32_t F, uint32_t full_levels, uint32_t split_permille,
                            int dist, uint32_t* node_lines, size_t cap_lines, uint64_t* tree_first_line);

/* ---- wire-format helpers (A2 packing, PipelinedMUX.sv:65) ------------------------------------ */
uint32_t orc_weights_lines_per_tree(uint32_t num_levels); /* ceil((2^(D+1)-1)/4) */
uint32_t orc_findex_lines_per_tree(uint32_t num_levels);  /* ceil((2^D-1)/8)     */
uint32_t orc_tuple_lines(uint32_t num_features);          /* ceil(F/4)           */

/* Build the two model streams from plain arrays.  thr/fidx/miss_right: [T][2^D-1] heap order,
 * leaves: [T][2^D].  wlines: T*wlpt*4 words, flines: T*flpt*8 u16, both zero padded. */
void orc_pack_model(uint32_t T, uint32_t D, const uint32_t* thr_bits, const uint16_t* fidx,
                    const uint8_t* miss_right, const uint32_t* leaf_bits, uint32_t* wlines,
                    uint16_t* flines);

/* ---- deterministic synthetic inputs --------------------------------- */
uint64_t orc_splitmix64(uint64_t x);
/* rows [row0, row0+n) of the benchmark tuple matrix, as tuple lines (ceil(F/4)*4 words per row).
 * dist 0: uniform [0,1) no missing (benchmark); dist 1: uniform [-1,1) with ~5% missing_bits. */
void orc_gen_tuples(uint64_t row0, size_t n, uint32_t F, int dist, uint32_t missing_bits, uint32_t* out);
/* synthetic model streams; dist as above (dist 1: thresholds in [-1,1)). */
void orc_gen_model(uint32_t T, uint32_t D, uint32_t F, int dist, uint32_t* wlines, uint16_t* flines);

int orc_hw_threads(void);

#ifdef __cplusplus
}
#endif
#endif
