/*
 * ddt.h -- C-ABI of libddt.so: MI355X-native decision-tree-ensemble scoring engine.
 *
 * The reference (fpgasystems/Distributed-DecisionTrees) is FPGA RTL with NO host software: the only
 * host-visible contract it defines is (1) the soft-register (CSR) parameter map,
 * rtl/DTEngine/EngineCSR.sv:190-305, and (2) the 128-bit-line stream formats for weights, feature
 * indexes, tuples and results, rtl/DTEngine/PCIeReceiver.sv:136-149 / ResultsCombiner.sv:136-203 with
 * the little-endian word packing of rtl/DTEngine/core/PipelinedMUX.sv:65.  This header is that contract
 * restated as a C-ABI: every entry point names the reference interface it replaces.  The function
 * names are this repository's (there is no reference software API to copy).
 *
 * One engine == one GPU (the analogue of one FPGA running DTInference, rtl/DTEngine/DTInference.sv:46-75).
 * Like the reference (one job between `start` and `process_done`, Core.sv:167-187) an engine is NOT
 * re-entrant: one in-flight ddt_score* per engine; independent engines are independent.
 * Ownership: the caller owns every buffer; the model is copied at load; tuple/score pointers are not
 * retained after a synchronous call returns (after stream completion for the *_device calls).
 *
 * There is NO CPU fallback in this library: without a usable HIP device ddt_create() fails.
 */
#ifndef DDT_H
#define DDT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDT_ABI_VERSION 6 /* 6: ddt_info::build_checks (was reserved), "sparse_r_*" kernels + option sparse_r32, ddt_debug_rank32_tables; 5: ddt_info grew (fallback_kernel); 4: hybrid jobs (ddt_comm_create_hybrid, ddt_score_hybrid_device, ...), ddt_comm_abort; 3: ddt_stats grew */

/* Threading: an engine is not thread-safe -- calls on ONE engine must not overlap; different engines (also on the
 * same device) are independent.  ddt_*_device calls are asynchronous on the given stream; ddt_destroy and
 * ddt_load_model* wait for the device before freeing or replacing what such work may still read. */

/* Return codes.  The reference has no error signalling beyond status counters (EngineCSR.sv:113-126);
 * every check below is an addition of this library. */
enum {
  DDT_OK = 0,
  DDT_EINVAL = -1,       /* bad argument / parameter out of range                          */
  DDT_ENOMEM = -2,       /* host or device allocation failed                               */
  DDT_EHIP = -3,         /* a HIP runtime call failed (see ddt_last_error)                 */
  DDT_ESTATE = -4,       /* call order: no model loaded, etc.                              */
  DDT_EUNSUPPORTED = -5, /* valid in the reference format but not supported by this build  */
  DDT_ENODEVICE = -6     /* no usable gfx950 device: the engine never falls back to CPU    */
};

/* Run parameters: one-to-one with the reference's CSR 204/205 fields (EngineCSR.sv:218-233).       */
typedef struct ddt_params {
  uint32_t num_trees;              /* trees in the model stream; slots beyond are EMPTY = +0 (DTPU.sv:544,760) */
  uint32_t num_levels;             /* D, compare levels per tree, CSR205[35:32]; 1..16                 */
  uint32_t num_features;           /* F; tuple lines = ceil(F/4) = CSR204[63:48]; 1..2048 (DTPU.sv:72)  */
  uint32_t missing_bits;           /* CSR205[31:0]: bit pattern that means "feature missing" (DTPU.sv:653) */
  uint32_t weights_lines_per_tree; /* CSR204[31:16]; >= ceil((2^(D+1)-1)/4)                            */
  uint32_t findex_lines_per_tree;  /* CSR204[47:32]; >= ceil((2^D-1)/8)                                */
  uint32_t cmp_mode;               /* 0 = reference compare: signed-int32 on raw fp32 bits (DTPU.sv:655);
                                      1 = IEEE-754 '<' (extension)                                      */
  uint32_t clusters_per_tuple;     /* C in {1,2,4,8}, CSR205[47:44]: fixes the fp32 summation order
                                      (Core.sv:291-316,486-541)                                         */
  uint32_t sum_mode;               /* 0 = the reference's adder network ORDER (FPAddersReduceTree.sv:94-141,
                                      FPAggregator.v:79-131, Core.sv:486-541) with IEEE-754 fp32 adds.  On the leaf
                                      domain the loader enforces (+0 or normal, 2^-102 <= |leaf| < 2^96; option
                                      leaf_domain_check) this equals the RTL's FloPoCo adder bit for bit EXCEPT in one
                                      case: an effective subtraction whose larger operand is an exact power of two,
                                      exponents exactly 25 apart, smaller mantissa != 0 -- the RTL returns the larger
                                      operand unchanged (FPAdder_2cycles_latency.v:325-326), IEEE the float just below
                                      it (1 ulp; e.g. 2^-4 + -1.5*2^-29: RTL 0x3D800000, IEEE 0x3D7FFFFF);
                                      2 = the same order with the reference adder itself (that case reproduced):
                                      bit-exact with the RTL's adder network on that domain, ~1 extra VALU op per add;
                                      1 = fp64 accumulate in stream order, rounded once to fp32 (extension)        */
  uint32_t reserved[3];            /* must be 0 */
} ddt_params;

typedef struct ddt_engine ddt_engine;

/* What the engine decided at load time; also the inputs of the roofline arithmetic. */
typedef struct ddt_info {
  uint32_t abi_version;
  int32_t  device_id;
  uint32_t tree_begin, tree_end;     /* shard held by this engine (global tree ids)                 */
  uint32_t num_levels, num_features, tuple_words; /* tuple_words = 4*ceil(F/4)                      */
  uint32_t variant;                  /* kernel variant id in use                                    */
  uint32_t tile_tuples, block_threads, lds_bytes;
  uint32_t num_classes, local_trees; /* classes (1 = plain ensemble); trees held by this engine      */
  uint64_t model_bytes_unpadded;     /* T_local*(4*(2^(D+1)-1) + 2*(2^D-1)): algorithmic model bytes */
  uint64_t image_bytes;              /* packed device image actually read per tile pass             */
  char     variant_name[64];
  char     device_name[64];
  uint32_t num_cus, clock_khz;       /* hipDeviceProp_t: multiProcessorCount, clockRate (inputs of the LDS / VMEM ceilings) */
  uint32_t lds_bytes_per_cu;
  uint32_t prepass_groups;           /* rank-quantised path: feature groups of the LDS-resident rank pre-pass (1 = all tables
                                        resident together, 2/4/8 = one launch split over groups), 0 = transpose + rank kernels
                                        or not the rank-quantised path                                                  */
  uint32_t fallback_kernel;          /* 1 = the model landed on a CORRECTNESS kernel ("generic" for perfect trees, "sparse_gf_*" for
                                        sparse forests): right results, no tuned path for this shape (depth / tuple width).  Callers
                                        that care about throughput should say so loudly (ddt_cli and bench.py do).              */
  uint32_t build_checks;             /* what the BUILD verified on this very binary's device code (csrc/ddt_checks.cpp): bit 0 = tools/check_s2_isa.py
                                        passed (no instruction touches the "_s2" kernels' SGPR record sets with their scalar loads in flight), bit 1 =
                                        tools/check_dma_waits.py passed (every counted s_waitcnt vmcnt(N) a barrier relies on covers the chunk DMA).
                                        A clear bit = that check could not run on the build machine: the automatic kernel choice then avoids the
                                        kernels concerned ("_s2" / "q16d_*"; environment DDT_DISABLE_S2=0 / DDT_DISABLE_DEEP=0 opts back in)      */
} ddt_info;

/* Observability counters, the analogue of CSR 220-226 / appStatus (EngineCSR.sv:113-126,
 * DTInference.sv:314-374). */
typedef struct ddt_stats {
  uint64_t tuples_in, tuples_out;    /* DTInference.sv:367-372                                      */
  uint64_t tuple_lines_in, result_lines_out, model_lines_in;
  uint64_t score_calls, kernel_launches;
  double   prog_ms, exec_ms;         /* progCycles / execCycles equivalents (host wall, ms)          */
  /* with ddt_set_option(e, "kernel_timing", 1): HIP-event times of the ddt_score_device launches on their stream -- pre-pass (rank
   * kernels, 0 for the fp32 kernels) and scoring kernel: of the LAST launch, and summed over all `timed_launches` so far.  The library
   * keeps the events of up to 64 launches; ddt_get_stats folds them in (it waits for the newest), so a loop of calls needs no host
   * synchronisation per call -- read the sums before and after it. */
  double   last_prepass_ms, last_score_ms;
  uint64_t timed_launches;
  double   sum_prepass_ms, sum_score_ms;
} ddt_stats;

/* -- lifecycle (replaces: CSR 200 start / reset, EngineCSR.sv:191-193, Core.sv:167-187) -------------- */
int  ddt_create(ddt_engine** out, int device_id);
void ddt_destroy(ddt_engine* e);

/* -- model load (replaces: the weights + feature-index PCIe streams, PCIeReceiver.sv:136-139,230-275,
 *    DTPU.sv:282-354, and CSR 202-205).  Both streams are in the reference wire format, host memory. -- */
int ddt_load_model(ddt_engine* e, const ddt_params* p, const void* weights_lines, size_t n_wlines,
                   const void* findex_lines, size_t n_flines);
/* Tree-sharded multi-device mode: keep only shard `shard_index` of `shard_count` contiguous shards of
 * ceil(T/shard_count) trees -- the per-device split of CSR 203 / PCIeReceiver.sv:241-264.            */
int ddt_load_model_shard(ddt_engine* e, const ddt_params* p, const void* weights_lines, size_t n_wlines,
                         const void* findex_lines, size_t n_flines, uint32_t shard_index,
                         uint32_t shard_count);
/* The split itself (host-only, needs no engine): shard `shard_index` of a list of `num_trees` trees is the contiguous
 * range [*tree_begin, *tree_end) of ceil(num_trees/shard_count) trees; trailing shards may be empty (begin == end).  It is
 * what the host node's stream router does with the model stream (PCIeReceiver.sv:241-264: `numcls_local_weights` lines per
 * device in list order; tests/test_oracle_receiver.py holds it to that block EXECUTED from the reference's source).  For a
 * multi-class model the list is the trees of ONE class.                                                              */
int ddt_shard_range(uint32_t num_trees, uint32_t shard_index, uint32_t shard_count, uint32_t* tree_begin,
                    uint32_t* tree_end);

/* -- scoring (replaces: the tuple PCIe stream in / result stream out, PCIeReceiver.sv:276-312,
 *    ResultsCombiner.sv:136-160,193; N need not be a multiple of 4 here, unlike A14) ------------------ */
/* -- pinned host buffers (replaces: the host-side DMA "slots" of the reference's PCIe path, PCIeShim.sv:99-130 -- the host
 *    program of the reference hands the FPGA buffers the DMA engine can read; pageable memory has to be staged first).
 *    ddt_host_register pins [ptr, ptr + bytes) for this engine's device (hipHostRegister); ddt_score / ddt_classify calls whose
 *    tuple buffer (and, for ddt_score, score buffer) lies inside a registered range then move the data with the DMA engine
 *    directly -- no staging copy through the engine's own pinned buffers.  Unregister before freeing the memory.          -- */
int ddt_host_register(ddt_engine* e, void* ptr, size_t bytes);
int ddt_host_unregister(ddt_engine* e, void* ptr);

/* Host buffers: tuple_lines = n_tuples * ceil(F/4) lines of 16 B; scores_out = n_tuples fp32.
 * Streams the batch through the pinned hipMemcpyAsync feeder (the PCIe feeder): three slots -- staging of chunk k+2 by host
 * threads, the link transfer of k+1 on one ordered copy stream, kernels + scores back of k; ranges pinned with ddt_host_register
 * skip the staging copies.                                                                           */
int ddt_score(ddt_engine* e, const void* tuple_lines, size_t n_tuples, float* scores_out);
/* Device buffers, asynchronous on `hip_stream` (a hipStream_t; NULL = the null stream).
 * Stream capture: once a call of at least this size has sized the engine's workspaces ("reserve_rows", or one warm-up call), the call only
 * enqueues work on `hip_stream` -- the tile flags' memset, the rank pre-pass, the scoring kernel(s), the combine of a cut launch -- with no
 * allocation and no synchronisation, and everything that depends on the batch (missing-value flags, partial sums, ticket counters) is rebuilt on
 * the stream: it may be captured into a HIP graph and replayed on new data in the same buffers (tests/test_graph_capture.py; not with
 * "kernel_timing").  A replay costs what the call costs on the device (profiles/r06_small_batches.md: the host's share drops from ~15 to ~8 us,
 * the call's latency does not move -- it is the device's).                                              */
int ddt_score_device(ddt_engine* e, const void* d_tuple_lines, size_t n_tuples, float* d_scores,
                     void* hip_stream);

/* -- multi-device combine (replaces: ResultsCombiner.sv:292-311,359-369 chain add) -------------------
 * d_parts = n_parts arrays of n fp32 partial scores laid out [part][n]; out[i] = (((p0+p1)+p2)+...),
 * i.e. the reference's host -> dev1 -> ... chain order.  Used by the deterministic multi-GPU path.   */
int ddt_chain_sum_device(ddt_engine* e, const float* d_parts, uint32_t n_parts, size_t n, float* d_out,
                         void* hip_stream);

/* -- multi-class one-vs-all (BASELINE config 5; an EXTENSION: the reference scores one fp32 sum per tuple and
 *    has no classes).  The model stream holds num_trees trees of num_classes classes; tree i belongs to class
 *    i % num_classes when `interleaved` != 0 (XGBoost multi:softprob order), else to class
 *    i / (num_trees / num_classes) (class-major; num_trees must then be a multiple of num_classes).  Every class
 *    is scored as an independent ensemble in the reference's adder order (ddt_params.clusters_per_tuple applies
 *    per class) and the label is argmax over classes, lowest index on ties.  With shard_count > 1 each class is
 *    tree-sharded like ddt_load_model_shard and the per-class partial sums must be combined across devices
 *    before the argmax (ddt_argmax_device).                                                                 -- */
int ddt_load_model_multiclass(ddt_engine* e, const ddt_params* p, const void* weights_lines, size_t n_wlines,
                              const void* findex_lines, size_t n_flines, uint32_t num_classes, int interleaved,
                              uint32_t shard_index, uint32_t shard_count);
/* d_class_scores: [num_classes][n] fp32 (this shard's partial sums per class), d_labels (may be NULL when
 * shard_count > 1): int32 argmax.  Asynchronous on hip_stream.                                               */
int ddt_classify_device(ddt_engine* e, const void* d_tuple_lines, size_t n_tuples, float* d_class_scores,
                        int32_t* d_labels, void* hip_stream);
/* host buffers; class_scores may be NULL */
int ddt_classify(ddt_engine* e, const void* tuple_lines, size_t n_tuples, int32_t* labels, float* class_scores);
int ddt_argmax_device(ddt_engine* e, const float* d_class_scores, uint32_t num_classes, size_t n, int32_t* d_labels,
                      void* hip_stream);

/* -- sparse (explicit-children) model stream (BASELINE config 4: deep random forests; an EXTENSION -- the reference
 *    only takes perfect trees of <= 16 levels whose 2^(D+1)-1 words fit a PU's BRAM, DTPU.sv:20-28; its own hook for
 *    bigger trees is the disabled hybrid path: entry bit 14 "next node is a leaf", DTPU.sv:637,661,675,712-715, and
 *    the PartialTrees control bit, Core.sv:380 bit 8, DTPU.sv:736-745 -- ill-defined in the published RTL, SURVEY
 *    A10b).  One 128-bit line per INTERNAL node, A2 packing (PipelinedMUX.sv:65):
 *      word 0        threshold bits
 *      word 1[15:0]  feature-index entry in the reference's bit layout (DTPU.sv:628,637,659-661): [10:0] feature,
 *                    [13] missing goes right, [14] the LEFT child is a leaf, [15] the RIGHT child is a leaf
 *                    (bit 14 keeps its RTL meaning "the next node is a leaf"); word 1[31:16] must be 0
 *      word 2, 3     left / right child: node index relative to the tree's first line, or the leaf's fp32 bits
 *    Children must have larger indices than their parent (BFS, DFS pre-order, ...) and the lines of a tree must form a
 *    TREE: every node but the root is the child of exactly one earlier node (a shared child or an unreachable node is
 *    DDT_EINVAL).  tree_first_line[num_trees + 1] delimits the trees; a tree that is a single leaf is one line with
 *    both leaf flags set and the value twice.  Width: the tuned sparse kernels keep a feature tile of 64..1024 tuples in
 *    LDS next to their top images, which takes tuples of up to ~540 words (tuple words * 256 B + 24 KiB <= 160 KiB); wider
 *    sparse models (num_features up to 2048) run on "sparse_gf_k6_u8_t256", which gathers every feature from the tuple's row in
 *    global memory -- the correctness path of this format, like the generic kernel of the perfect-tree format.
 *    ddt_params: num_levels = upper bound of the depth (1..64), weights/findex lines per tree ignored.  Compare rule,
 *    missing rule, EMPTY slots, summation order and tree sharding are exactly those of the perfect format; the model
 *    is then scored with ddt_score / ddt_score_device.                                                           -- */
int ddt_load_model_sparse(ddt_engine* e, const ddt_params* p, const void* node_lines, size_t n_lines,
                          const uint64_t* tree_first_line, uint32_t shard_index, uint32_t shard_count);
/* one-vs-all classes in a sparse stream (random-forest classifiers): class membership, sharding and argmax as in
 * ddt_load_model_multiclass; score with ddt_classify* / ddt_classify_sharded_device */
int ddt_load_model_sparse_multiclass(ddt_engine* e, const ddt_params* p, const void* node_lines, size_t n_lines,
                                     const uint64_t* tree_first_line, uint32_t num_classes, int interleaved,
                                     uint32_t shard_index, uint32_t shard_count);

/* -- multi-GPU jobs: RCCL over xGMI behind the C-ABI ---------------------------------------------------------------
 *    Replaces the inter-FPGA networks of the reference: the ring broadcast of tuple lines (InputDistributor.sv:199-204)
 *    and the partial-result network (ResultsCombiner.sv:292-311 chain adders, :359-369,426-430 forwarding).  Both of
 *    the reference's modes (DTInference.sv:28-37):
 *      tree-sharded  rank g holds shard g of ceil(T/G) contiguous trees (ddt_load_model_shard / _sparse / _multiclass
 *                    with shard_index = rank, shard_count = n_ranks), every rank scores ALL tuples, the per-tuple
 *                    partial scores are combined across ranks, chunk-pipelined (chunk k's collective runs on the
 *                    comm's own stream while chunk k+1 is being scored):
 *                      DDT_COMBINE_ALLREDUCE  one ncclAllReduce(ncclFloat, ncclSum) per chunk; the ring's summation
 *                                             order is RCCL's: equal to the reference's chain order to fp32 rounding;
 *                      DDT_COMBINE_CHAIN      deterministic: all-to-all (grouped ncclSend / ncclRecv of 1/G slices),
 *                                             fixed-order add p0 + p1 + ... on the owner (ResultsCombiner.sv:292-311's
 *                                             host -> dev1 -> ... order), ncclAllGather: bit-exact with the chain.
 *      row-sharded   every rank holds the whole ensemble and scores rows [r*ceil(n/G), ...) in place in the caller's buffer,
 *                    "chunk_rows" / G rows at a time; while the next step is being scored the comm stream hands the finished one
 *                    to every peer with grouped ncclSend / ncclRecv straight into its place in their buffers (one message per
 *                    xGMI link at once; exact per-peer counts: no padding, no staging) -- every rank ends up with the full
 *                    score vector ("replicas only": no arithmetic crosses devices).
 *    One ddt_comm per (process or thread) x device; the calls are collective: every rank calls them with the same
 *    arguments (tuples replicated on every rank for the tree-sharded calls).  They are asynchronous on `hip_stream`
 *    like ddt_score_device; workspace buffers are (re)allocated synchronously when a call needs more than before.
 *    Multi-process (one process per GPU): rank 0 calls ddt_comm_get_unique_id and hands the 128 bytes to every rank
 *    (any launcher-side channel: a file, MPI, torch.distributed's store), then every rank calls ddt_comm_create.
 *    Single process driving several GPUs: ddt_group_* below (ncclCommInitAll + one worker thread per device).
 *    An engine that has been given a communicator of more than one rank stays in "job mode" for its lifetime (its automatic kernel
 *    choice prefers the persistent depth-8 kernel, whose blocks do not wait for CUs that collective kernels occupy): the communicator
 *    may outlive or predecease the engine, so destroying it does not touch the engine; use a fresh engine for stand-alone scoring.  -- */
typedef struct ddt_comm ddt_comm;
#define DDT_COMM_ID_BYTES 128
enum { DDT_COMBINE_ALLREDUCE = 0, DDT_COMBINE_CHAIN = 1 };
int  ddt_comm_get_unique_id(void* id_out /* DDT_COMM_ID_BYTES */);
int  ddt_comm_create(ddt_comm** out, ddt_engine* e, int rank, int n_ranks, const void* unique_id);
void ddt_comm_destroy(ddt_comm* c);
const char* ddt_comm_last_error(const ddt_comm* c);
/* "chunk_rows": rows per pipelined collective (default 12,500,000); "taper_tail": 1 = the last chunk is cut into 1/2, 1/4, 1/4
 * (whole 1024-tuple tiles, pieces of at least "taper_min_rows" = 2^20 rows) so that the collective left exposed behind the last
 * scoring launch is a quarter of the size, 0 = never, -1 (default) = when the communicator has more than one rank;
 * "comm_stream_priority": 1 = run the collectives on a stream of the device's highest priority (default 0) */
int  ddt_comm_set_option(ddt_comm* c, const char* key, int64_t value);
/* the chunk lengths a sharded call of n_tuples rows uses (host-only; every rank derives the same list): returns their number,
 * lens_out (may be NULL to count) receives them */
int64_t ddt_comm_chunk_schedule(size_t n_tuples, size_t chunk_rows, int taper, size_t taper_min_rows, size_t* lens_out,
                                size_t cap);
int  ddt_score_sharded_device(ddt_comm* c, const void* d_tuple_lines, size_t n_tuples, float* d_scores, int combine,
                              void* hip_stream);
int  ddt_score_rowsharded_device(ddt_comm* c, const void* d_tuple_lines, size_t n_tuples, float* d_scores,
                                 void* hip_stream);
/* host buffers (the per-rank counterpart of ddt_group_score; `ddt_cli score --ranks N --rank r` is built on it): the tuples
 * go to this rank's device in super-chunks ("host_rows" option, default 2^23 rows), through ddt_score_sharded_device, and the
 * combined scores come back to every rank.  With peers the host tuples cross PCIe ONCE: rank r copies 1/n of a super-chunk and
 * every rank hands its rows to all peers over xGMI (the reference re-broadcasts tuple lines along its ring,
 * InputDistributor.sv:199-204); option "tuple_broadcast" = 0 makes every rank copy everything from the host instead.  The same
 * holds for ddt_group_score / ddt_group_classify.  Collective and synchronous. */
int  ddt_comm_score(ddt_comm* c, const void* tuple_lines, size_t n_tuples, float* scores_out, int combine);
/* multi-class (ddt_load_model_multiclass with shard_index = rank): per-class partial sums [K][n] combined like the
 * scalar scores, then the argmax over the combined sums (d_labels may be NULL) */
int  ddt_classify_sharded_device(ddt_comm* c, const void* d_tuple_lines, size_t n_tuples, float* d_class_scores,
                                 int32_t* d_labels, int combine, void* hip_stream);

/* -- the two modes composed ("hybrid": tree groups x row groups) -----------------------------------------------------------
 *    The reference runs either mode (DTInference.sv:28-37): the ensemble spread over the devices and the partial results added along
 *    the chain (PCIeReceiver.sv:241-264, ResultsCombiner.sv:292-311), or the ensemble on every device and the tuples dealt out, results
 *    interleaved (PCIeReceiver.sv:289-312, ResultsCombiner.sv:371-391).  On a node of 8 GPUs the composition is the fast one: the n
 *    ranks form n / tree_ranks ROW GROUPS of tree_ranks consecutive ranks; rank r holds tree shard r % tree_ranks of tree_ranks
 *    (ddt_load_model_shard / _sparse / _multiclass with exactly that shard_index / shard_count) and its row group r / tree_ranks scores
 *    rows [lo, hi) = ddt_hybrid_rows(n, row groups, row group) of a batch.  Inside a row group the job is the tree-sharded one (chunk
 *    pipeline, all-reduce or chain, on a communicator split off the world communicator with ncclCommSplit); the work a tree-sharded
 *    rank does for EVERY tuple (reading it, ranking it against the shard's thresholds) shrinks by the number of row groups, and the
 *    all-reduce spans tree_ranks ranks.  tree_ranks == n_ranks is the tree-sharded job, tree_ranks == 1 the row-sharded replicas.
 *    gather != 0: while the next piece is scored, every finished piece is handed to the ranks with the same tree shard in the other row
 *    groups (grouped ncclSend / ncclRecv on the world communicator, straight into place): every rank ends up with all n rows, as in the
 *    other two jobs.  gather == 0: a rank only writes rows [lo, hi) of the result (scores stay with their row group, the way the
 *    reference returns a device's rows from that device).  Only rows [lo, hi) of d_tuple_lines are read.
 *    ddt_comm_score on a hybrid communicator: host tuples cross PCIe once in the whole job (1 / n_ranks per rank, handed on inside the
 *    row group over xGMI), every rank receives all scores.  ddt_score_sharded_device / ddt_classify_sharded_device /
 *    ddt_score_rowsharded_device refuse a hybrid communicator (DDT_ESTATE), and the hybrid calls refuse a plain one.               -- */
typedef struct ddt_comm_layout_t {
  int rank, n_ranks;           /* in the world communicator */
  int tree_ranks, tree_rank;   /* size of / rank inside the communicator the partial scores are combined over = shard_count / shard_index
                                  this rank's engine must hold (plain communicator: n_ranks, rank) */
  int row_groups, row_group;   /* plain communicator: 1, 0 */
} ddt_comm_layout_t;
int  ddt_comm_create_hybrid(ddt_comm** out, ddt_engine* e, int rank, int n_ranks, int tree_ranks, const void* unique_id);
int  ddt_comm_layout(const ddt_comm* c, ddt_comm_layout_t* out);
/* host-only: rows [*lo, *hi) of row group `row_group` of `row_groups`: equal slices in whole 1024-tuple tiles (whole result lines), the
 * last group takes what is left, groups past the end are empty */
int  ddt_hybrid_rows(size_t n_tuples, int row_groups, int row_group, size_t* lo, size_t* hi);
int  ddt_score_hybrid_device(ddt_comm* c, const void* d_tuple_lines, size_t n_tuples, float* d_scores, int combine, int gather,
                             void* hip_stream);
int  ddt_classify_hybrid_device(ddt_comm* c, const void* d_tuple_lines, size_t n_tuples, float* d_class_scores, int32_t* d_labels,
                                int combine, int gather, void* hip_stream);
/* A peer failed before (or inside) its collective -- its call returned an error, or its process died: the other ranks' streams would
 * wait for it for ever (RCCL has no timeout of its own).  The launcher tells the survivors, they call ddt_comm_abort (ncclCommAbort on
 * the communicators): their queued collectives end, streams and ddt_comm_destroy return; the results of the job are undefined and the
 * communicator refuses every further call with DDT_ESTATE.  (The reference has no such path: a stalled device stalls the ring.) */
int  ddt_comm_abort(ddt_comm* c);

/* Single-process multi-GPU job: n engines + one RCCL communicator over them (ncclCommInitAll), one worker thread per
 * device.  ddt_group_load_model* gives device g tree shard g; ddt_group_score takes HOST buffers, replicates the
 * tuples on every device (one H2D copy per device, the analogue of the reference's tuple broadcast), runs the
 * tree-sharded job and returns the combined scores from device 0.  ddt_group_engine(g, i) exposes engine i
 * (options, stats). */
typedef struct ddt_group ddt_group;
int  ddt_group_create(ddt_group** out, int n_devices, const int* device_ids);
/* the hybrid layout in one process: row groups of tree_ranks consecutive devices (tree_ranks divides n_devices);
 * ddt_group_load_model* gives device i tree shard i % tree_ranks of tree_ranks, ddt_group_score / ddt_group_classify run the hybrid job
 * (tuples over PCIe once, 1 / n_devices per device) and return all rows from device 0 */
int  ddt_group_create_hybrid(ddt_group** out, int n_devices, const int* device_ids, int tree_ranks);
void ddt_group_destroy(ddt_group* g);
const char* ddt_group_last_error(const ddt_group* g);
ddt_engine* ddt_group_engine(ddt_group* g, int index);
int  ddt_group_load_model(ddt_group* g, const ddt_params* p, const void* weights_lines, size_t n_wlines,
                          const void* findex_lines, size_t n_flines);
int  ddt_group_load_model_sparse(ddt_group* g, const ddt_params* p, const void* node_lines, size_t n_lines,
                                 const uint64_t* tree_first_line);
int  ddt_group_score(ddt_group* g, const void* tuple_lines, size_t n_tuples, float* scores_out, int combine);
/* the reference's OTHER mode in one process (DTInference.sv:33-36: the ensemble fits one device, the tuples are partitioned):
 * every device loads the whole ensemble; device d scores rows [d * per, ...) of the host buffer through its own engine's feeder
 * (pinned double buffer over ITS PCIe link) and writes them to their place in scores_out -- no collective at all */
int  ddt_group_load_model_replicated(ddt_group* g, const ddt_params* p, const void* weights_lines, size_t n_wlines,
                                     const void* findex_lines, size_t n_flines);
int  ddt_group_score_rows(ddt_group* g, const void* tuple_lines, size_t n_tuples, float* scores_out);
/* multi-class models (config 5): class k's trees sharded tree-wise like the scalar ensemble; labels_out [n] int32 argmax of
 * the combined per-class sums, class_scores_out (may be NULL) [num_classes][n] */
int  ddt_group_load_model_multiclass(ddt_group* g, const ddt_params* p, const void* weights_lines, size_t n_wlines,
                                     const void* findex_lines, size_t n_flines, uint32_t num_classes, int interleaved);
int  ddt_group_classify(ddt_group* g, const void* tuple_lines, size_t n_tuples, int32_t* labels_out, float* class_scores_out,
                        int combine);

/* -- introspection ----------------------------------------------------------------------------------- */
int ddt_get_info(const ddt_engine* e, ddt_info* out);
int ddt_get_stats(const ddt_engine* e, ddt_stats* out);
const char* ddt_strerror(int code);
const char* ddt_last_error(const ddt_engine* e); /* detail of the last failure on this engine */
/* Tuning knobs: "variant" (kernel variant id, -1 = auto; a loaded model is re-packed -- a variant that does not fit it is refused with
 * DDT_EUNSUPPORTED and the model stays loaded as it was), "feeder_rows" (rows per feeder
 * chunk, default 2^20; a chunk never holds more than 512 MiB of tuples), "feeder_threads" (host threads of the staging copy, default 8), "kernel_timing" (see ddt_stats),
 * "q16_fused_prepass" / "q16_grouped_prepass" (1 = default; 0 = never rank with all tables resident together / never
 * split the rank pre-pass over feature groups; both 0 = the transpose + rank kernels) and "q16_prepass_groups" (0 =
 * cheapest, default; 1, 2, 4, 8 = exactly that many feature groups): A/B switches, effective at the next model load;
 * "q16_max_table" (255..38848, default 38848: the distinct thresholds per feature ONE rank table may hold -- what fits a block's LDS in the rank
 * kernel; an ensemble beyond it is scored in parts; 32767 = the limit until round 6; effective at the next model load);
 * "q16_persistent" (-1 = default: the persistent depth-8 rank-quantised kernel -- resident blocks that take tiles from a ticket
 * counter -- where it wins: one-vs-all models whose classes hold equally many trees, scored in ONE launch, and engines inside a
 * multi-rank job; 1 = wherever it fits; 0 = never), "q16_prepass_nt" (A/B: bit 0 / 1 = nontemporal stores / loads in the rank
 * pre-pass; measured no gain, default 0), "q16_walk_padding" (A/B: 0 = default, the plain rank-quantised kernels do not walk the
 * EMPTY trees that pad the image to whole chunks -- their +0 leaves are added as always, the scores are the same bit for bit; 1 =
 * walk them, as before round 4);
 * (environment DDT_DEBUG_PREPASS=1 prints the chosen plan -- groups, P, image bytes -- to stderr at load);
 * "class_streams" (1 = default: the per-class scoring launches of a multi-class model alternate between the caller's stream and
 * one stream of the engine, joined before the argmax / before the call's work is visible on the caller's stream; 0 = one stream);
 * "stream_blocks_per_cu" (0 = default: the persistent stream kernel launches as many blocks per CU as are resident; 1..16
 * forces the number: A/B switch), "stream_res_tiles" / "stream_window_ticks" (the stream kernel's phased result stores: every
 * wave parks its scores in LDS and all waves of the chip write them in the same short window, so that the HBM serves an unmixed
 * read stream in between; 0 = default: as many score slots as the LDS leaves / a window every 30 us; stream_res_tiles 1 = direct
 * stores, n = at most n slots; stream_window_ticks in ticks of the device's wall clock -- hipDeviceAttributeWallClockRate, 100 MHz = 10 ns on MI355X), "reserve_rows" (pre-size the
 * workspace of the rank-quantised path for calls of up to that many rows: the *_device calls then never allocate),
 * "leaf_domain_check" (1 = default: refuse -0 / sub-normal / Inf / NaN leaves in the reference-order sum, where the
 * GPU's IEEE adds and the reference's adder differ), "sparse_top_levels" (-1 = auto, or 6..10 levels of a sparse
 * forest staged in LDS), "sparse_deep_order" (0 = level order, default; 1 = depth-first per sub-tree), "sparse_q16" (1 =
 * default: sparse forests whose distinct thresholds per feature fit 16-bit ranks -- at most 38848 (what one block of the rank kernel holds in LDS), e.g. histogram-trained
 * models -- and whose tuples have at most 64..76 words run on the rank-quantised sparse kernels: u16 feature tile, 1024
 * tuples per block; 0 = always the fp32-tile kernels), "sparse_dk" (1 = default: the "dense level K" sparse kernels where they
 * exist -- all top levels as 8-byte records in LDS, the first deep level addressed by the heap index; 0 = never: only the kernels with 16-byte
 * level K-1 records, which since round 6 exist for the 128- and 64-tuple tiles of wide tuples only -- an A/B and test switch), "sparse_r32"
 * (-1 = default: sparse forests of depth >= 13 with at least two trees per tuple word, tuples of at most 128 words and at most 131,068
 * distinct thresholds per feature run on the "sparse_r_*" kernels -- thresholds as
 * 17-bit ranks, a 32-bit rank tile written by a pre-pass per batch, 16-byte PAIR records {node, left child, right child, pointer}
 * on every level below the top image: one gather decides two levels; 0 = never; 1 = wherever such a kernel fits: at most 128 features and 131,070
 * distinct thresholds per feature).  A refused
 * sparse_* value keeps the previous one and the loaded model. */
int ddt_set_option(ddt_engine* e, const char* key, int64_t value);
int ddt_num_variants(void);
int ddt_variant_name(int variant, char* buf, size_t buflen);

/* -- soft-register (CSR) block codec: the reference's run parameters as the host writes them, CSR 200..211
 *    (EngineCSR.sv:190-305).  csr[k] is the 64-bit value written to register 200+k.  encode() fills the fields
 *    this engine consumes plus sane values for the FPGA-only ones (mode flags, packet sizes); decode() recovers
 *    ddt_params, the tuple count (4 * CSR207[31:0], the reference counts whole result lines) and the device
 *    count (CSR203[39:32]).  num_features comes back as 4 * tuple lines (the wire format does not carry F);
 *    cmp_mode / sum_mode are not in the CSR map and decode to 0 (the RTL's behaviour).                      -- */
#define DDT_CSR_FIRST 200
#define DDT_CSR_COUNT 12
int ddt_csr_encode(const ddt_params* p, uint64_t n_tuples, uint32_t num_devices, uint64_t csr[DDT_CSR_COUNT]);
int ddt_csr_decode(const uint64_t csr[DDT_CSR_COUNT], ddt_params* p, uint64_t* n_tuples, uint32_t* num_devices);
/* Per-device blocks of a multi-device job (EngineCSR.sv:194-205 mode flags, :235-243 next-hop addresses, :250-296
 * device list at a byte stride).  shard_mode: the reference's two modes (DTInference.sv:28-37) -- DDT_SHARD_TREES =
 * ensemble spread over the devices, tuples broadcast, partial results aggregated (broadcast_data + aggreg_enabled);
 * DDT_SHARD_ROWS = every device holds the whole ensemble, tuples dealt in batches of 4, results forwarded
 * (broadcast_trees).  device_index 0 is the host node, the last one sets last_node.  ddt_csr_encode() is
 * (DDT_SHARD_TREES, device 0).  decode_ex also returns the mode, the raw CSR201[7:0] flags and the 20 device ids
 * sliced the way the RTL slices them (any of the three may be NULL). */
enum { DDT_SHARD_TREES = 0, DDT_SHARD_ROWS = 1 };
int ddt_csr_encode_ex(const ddt_params* p, uint64_t n_tuples, uint32_t num_devices, uint32_t shard_mode,
                      uint32_t device_index, uint64_t csr[DDT_CSR_COUNT]);
int ddt_csr_decode_ex(const uint64_t csr[DDT_CSR_COUNT], ddt_params* p, uint64_t* n_tuples, uint32_t* num_devices,
                      uint32_t* shard_mode, uint32_t* mode_flags, uint8_t device_ids[20]);

/* -- deterministic synthetic inputs of SURVEY.md 8(d) (bench/test support; device generator so that
 *    the timed region starts with inputs resident in HBM) -------------------------------------------- */
int ddt_synth_model(uint32_t num_trees, uint32_t num_levels, uint32_t num_features, int dist,
                    void* weights_lines, void* findex_lines);
/* synthetic random-forest-like SPARSE model (deterministic): nodes above `full_levels` are all internal, below a
 * child is internal with probability split_permille / 1000 until max_depth.  Returns the number of node lines
 * (negative DDT_E* on bad arguments); call with node_lines == NULL to size the buffers first. */
int64_t ddt_synth_sparse_model(uint32_t num_trees, uint32_t max_depth, uint32_t num_features, uint32_t full_levels,
                               uint32_t split_permille, int dist, void* node_lines, size_t cap_lines,
                               uint64_t* tree_first_line);
int ddt_synth_tuples_host(void* tuple_lines, uint64_t row0, size_t n, uint32_t num_features, int dist,
                          uint32_t missing_bits);
int ddt_synth_tuples_device(ddt_engine* e, void* d_tuple_lines, uint64_t row0, size_t n,
                            uint32_t num_features, int dist, uint32_t missing_bits, void* hip_stream);

/* -- host-only test hook (needs no GPU): the LDS images of the rank pre-pass of the rank-quantised path -------------
 *    keys = the sorted (signed int32 order) distinct threshold keys of every feature, concatenated; counts[w] per tuple
 *    word w (n_words = 4 * ceil(F / 4) <= 32).  groups: 0 = the engine's choice, else 1 / 2 / 4 / 8 feature groups.
 *    plan_out[0] = groups built (0: tables too big, the engine would fall back to its transposed pre-pass), [1] = tuple
 *    lines per group, then per group g {image byte offset, image bytes, parameter-block byte offset, P, first line} at
 *    plan_out[2 + 5 g].  image_out (may be NULL to size it) receives the concatenated images; returns the number of
 *    32-bit words they take (negative DDT_E* on bad arguments).  Layout: DESIGN.md section 3; tests/test_prepass_host.py
 *    replays the kernel's search on it against a plain sorted-table count. */
int64_t ddt_debug_prepass_image(const uint32_t* keys, const uint32_t* counts, uint32_t n_words, uint32_t groups,
                                uint32_t* image_out, size_t image_cap_words, uint32_t plan_out[42]);
/*    the device image of a PERFECT-tree model as kernel variant `variant_id` wants it (-1 = the variant the engine would pick
 *    for this model: its id comes back in info_out[10]); the streams are validated and parsed like ddt_load_model does, then
 *    packed (csrc/ddt_internal.h).  info_out = {image words, trees incl. EMPTY padding, variant kind
 *    (0 generic, 1 tile, 2 stream, 3 rank-quantised), variant opt bits, trees per chunk, tuples per tile, LDS byte address of
 *    feature row 0, bytes per feature row, Kpad (rank-quantised: entries per table), tuple words, variant id, table words}.
 *    Rank-quantised variants also return the image used by tiles that hold a missing value (img_slow_out) and the per-feature
 *    sorted threshold tables [tuple words][Kpad] (tables_out).  All outputs may be NULL to size them first.
 *    tests/test_image_host.py walks the images in numpy the way the kernels do, against the oracle.                       */
int ddt_debug_model_image(const ddt_params* p, const void* weights_lines, size_t n_wlines, const void* findex_lines,
                          size_t n_flines, int variant_id, uint32_t* img_out, uint32_t* img_slow_out, size_t img_cap_words,
                          uint32_t* tables_out, size_t tables_cap_words, uint64_t info_out[12]);
/*    the device images of a sparse forest as sparse kernel variant `variant_id` wants them: the whole
 *    stream is validated like ddt_load_model_sparse does, then packed -- top heap images per PU group and the deep record
 *    array.  info_out = {top words, deep words, PU groups, top levels K, LDS byte offset of feature row 0, bytes per
 *    feature row}; top_out / deep_out may be NULL to size them.  tests/test_sparse_host.py walks the images in numpy. */
int ddt_debug_sparse_image(const ddt_params* p, const void* node_lines, size_t n_lines, const uint64_t* tree_first_line,
                           int variant_id, int deep_order, uint32_t* top_out, size_t top_cap_words, uint32_t* deep_out,
                           size_t deep_cap_words, uint64_t info_out[6]);

/*    ... for a "sparse_r_*" variant (32-bit ranks, pair records): top = one-word nodes, deep = pair / LEAF records (csrc/ddt_internal.h "32-bit
 *    ranks"); info_out[3] = K | (record hops on the longest path below the top image) << 32.
 *    The tables of the 32-bit rank pre-pass (csrc/ddt_sparse_r.hip rank32_kernel): keys / counts as for ddt_debug_prepass_image (n_words <= 2048,
 *    counts <= 2^17 - 2).  dir_out [n_words][Kpad] = the directory (last key of every block of 2^blk_log2 keys), par_out [n_words][8] = {directory
 *    entries, lo, hi, shift, P, key offset of the feature's blocks in tab_out, real keys, largest key}, starts_out [n_words][4096] bucket starts,
 *    tab_out = the key blocks.  info_out = {Kpad, blk_log2, tab words, 0}.  Outputs may be NULL to size them.  tests/test_sparse_r_host.py replays
 *    the kernel's search on them against a plain sorted-table count. */
int ddt_debug_rank32_tables(const uint32_t* keys, const uint32_t* counts, uint32_t n_words, uint32_t* dir_out, size_t dir_cap_words,
                            uint32_t* par_out, uint16_t* starts_out, uint32_t* tab_out, size_t tab_cap_words, uint64_t info_out[4]);

#ifdef __cplusplus
}
#endif
#endif /* DDT_H */
