// ddt_internal.h -- shared between the host side (ddt_engine.cpp, ddt_image.cpp, ddt_choice.cpp, ...) and the HIP kernels (ddt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/ddt.h"

namespace ddt {

// ---------------------------------------------------------------------------------------------------
// Packed tree image ("LDS image").  One tree occupies TREE_BYTES = 12 * 2^D bytes.
//   layout 0 (classic):
//     [0, 8*2^D)        node records, 1-BASED heap (record 0 is padding): {u32 thr_key, u32 w2}
//                       w2 = feature word (kernel specific, see build_image) | miss_right << 31
//     [8*2^D, 12*2^D)   2^D fp32 leaves, left to right
//   layout 1 (last level fused with its leaves, Variant::opt bit 0):
//     [0, 4*2^D)        records of levels 0..D-2 (1-based heap, record 0 padding)
//     [4*2^D, 12*2^D)   2^(D-1) records of 16 bytes {thr_key, w2, leaf_left, leaf_right} for level D-1
// Heap walk in byte units: m8 = 8 (root); m8 <- 2*m8 + 8*right.
// Reference semantics: rtl/DTEngine/core/DTPU.sv:579-760 (0-based n' = 2n+1+right is the same walk).
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kFlagMissRight = 0x80000000u;
constexpr uint32_t kMissSentinelIeee = 0x7FFFFFFEu;  // key-space missing marker for cmp_mode 1

struct ScoreArgs {
  const uint4* img;        // packed image: n_trees * tree_bytes
  const uint32_t* tuples;  // device, row-major tuple lines: tuple_words u32 per tuple
  float* out;              // device, one fp32 per tuple
  uint64_t n;              // tuples
  uint32_t tuple_words;    // 4*ceil(F/4)
  uint32_t n_trees;        // trees in the image (padded to the variant's granule with EMPTY trees)
  uint32_t n_chunks;       // tile kernels: image chunks
  uint32_t levels;         // D
  uint32_t clusters;       // C (reference summation order)
  uint32_t miss_raw;       // params.missing_bits (tested on raw tuple bits while staging)
  uint32_t miss_key;       // what a missing feature looks like inside LDS (== miss_raw for cmp_mode 0)
  uint32_t ieee;           // 1: stage features through the IEEE order-preserving key transform
  uint32_t sum_mode;       // 0 reference-order fp32, 1 fp64 sequential
  const void* aux;         // kind-specific extras (Q16Aux for the rank-quantised path), else NULL
  uint32_t top_levels;     // generic kernel, deep trees: levels of each tree staged in LDS (set by launch_generic)
  hipEvent_t ev_mid;       // optional ("kernel_timing"): recorded right before the scoring kernel proper
  uint32_t num_cus;        // hipDeviceProp_t::multiProcessorCount: sizes the persistent grids
  uint32_t stream_blocks_per_cu;  // stream kernel: 0 = as many blocks per CU as are resident, else forced (option, A/B)
  // stream kernel, phased result stores (ddt_kernels.hip).  From the engine: the options "stream_res_tiles" (0 = as many slots as
  // the LDS leaves, 1 = direct stores, n = at most n) and "stream_window_ticks" (0 = default); launch_stream() fills in what the
  // kernel reads: the slots per wave (0 = direct stores), the window in 10 ns ticks and the LDS byte offset of the slots.
  uint32_t stream_res_tiles;
  uint32_t stream_window_ticks;
  uint32_t stream_res_off;
};

constexpr uint32_t kQMissing = 0xFFFFu;  // rank of a missing feature value in the u16 tiles
constexpr uint32_t kQ16RankBuckets = 4096;
// LDS-resident rank pre-pass (no transposed fp32 intermediate): the features are cut into `groups` groups of `lines`
// tuple lines (4 features each); a block keeps the tables of ONE group resident in LDS (image = bytes[g] at byte offset
// img_off[g] of Q16Aux::prepass_img; layout: ddt_image.cpp build_prepass_group).  groups == 1: fused_rank_kernel (all
// tables fit together, e.g. a 125-tree shard).  groups > 1: grouped_rank_kernel, ONE launch whose blocks are split over
// the groups and over `parts` row partitions (= XCDs: the blocks of all groups that work on the same rows share one L2,
// so a tuple row leaves HBM once although every group reads it).
constexpr uint32_t kQ16MaxGroups = 8, kQ16Segments = 32;
struct PrepassPlan {
  uint32_t groups = 0, lines = 0;
  uint32_t img_off[kQ16MaxGroups] = {}, bytes[kQ16MaxGroups] = {}, par_off[kQ16MaxGroups] = {}, P[kQ16MaxGroups] = {};
  uint32_t line_lo[kQ16MaxGroups] = {};
};
constexpr uint32_t kQ16GroupedCounters = 64;  // 8-byte work counters behind the tile flags: one per (group, part)

struct Q16Aux {               // device pointers of the rank-quantised path (ScoreArgs::aux)
  uint32_t* xT;               // workspace [W][n_pad]: transposed tuples
  uint16_t* q;                // workspace [tiles][W][1024]: feature ranks
  uint32_t* tile_flags;       // workspace [tiles]: 1 = the tile holds a missing value
  const uint32_t* tables;     // [W][Kpad] sorted distinct threshold keys per feature, padded with INT_MAX
  const uint32_t* tabP;       // [W][8] per feature: K (real table length), lo, hi (first / last key), shift, P, 0, 0, 0
  const uint16_t* tabS;       // [W][kQ16RankBuckets] bucket starts: number of keys in the slices below (rank_kernel)
  uint32_t Kpad;              // entries per table, > max table length (ddt_engine_priv.h q16_table_pad: a power of two up to 32767 keys)
  const uint4* img_slow;      // image with the miss_right flags (used by tiles that contain a missing value)
  uint64_t n_pad;             // rows rounded up to whole tiles of 1024
  uint32_t skip_prepass;      // 1 = q / tile_flags already hold this batch (2nd..Kth class of a multi-class model)
  uint32_t skip_transpose = 0;  // transpose + rank pre-pass: xT already holds this batch's transposed tuples (a later part of an ensemble scored
                              // in parts: the tables differ from part to part, the transposed tuples do not)
  const uint4* prepass_img;   // LDS-resident pre-pass: concatenated LDS images, one per feature group
  PrepassPlan prepass;        // groups == 0: use transpose_kernel + rank_kernel
  uint32_t real_groups;       // "_cm" kernels: PU groups that hold a real tree, ceil(T / 8) (the image may be padded with EMPTY groups)
  // plain (non-persistent) kernels: sub-groups (of the kernel's U trees in flight) at the front of this launch's image that hold a
  // real tree; the rest -- EMPTY padding up to whole chunks, always at the image's end -- is not walked, its +0 leaves are added as
  // always (the result is the same bit for bit: an EMPTY tree's walk ends in a +0 leaf whatever the tuple).  0 = walk everything.
  uint32_t walk_subgroups = 0;
  unsigned long long* dbg = nullptr;  // "_p" kernels, environment DDT_Q16P_PROFILE: per block 8 x 64-bit phase sums in 10 ns ticks (launch_q16p prints them)
  // "_p" (persistent) kernels.  The image may hold SEVERAL ensembles back to back ("segments": the classes of a one-vs-all model,
  // each padded to seg_chunks whole chunks, real_groups real PU groups in each): segment k's sum goes to out[k * n + row] and the
  // label (argmax over the segments, lowest index on ties, a NaN never beats a number) to labels[row].  n_segs == 1: plain scores.
  uint32_t n_segs = 1;        // ensembles in the image
  uint32_t seg_chunks = 0;    // chunks per ensemble (0 = all of ScoreArgs::n_chunks)
  int32_t* labels = nullptr;  // n_segs > 1: argmax over the segments (may be NULL); `ScoreArgs::out` (may be NULL then) = [n_segs][n] sums
  // n_segs > 1, trees per ensemble mod 8 in 1..4: the second half of every ensemble's partly filled PU group (= its chunk's second
  // sub-group) is EMPTY padding; that sub-group's walk is skipped, its +0 leaves are added as always.  The value is what the kernel's
  // count-down of an ensemble's chunks (seg_chunks .. 1) reads AT that chunk: seg_chunks - (the chunk's index in the ensemble's image).  In a
  // cluster-major image the partial group is the last of ITS cluster's run, not necessarily of the image (ddt_image.cpp cm_position).
  // 0 = nothing to skip (the count-down never reads 0).
  uint32_t seg_tail_left = 0;
  uint32_t* tile_counter = nullptr;  // work counter of the persistent blocks (zeroed per launch; engine workspace behind the pre-pass counters)
  uint32_t prepass_nt = 0;    // A/B (option "q16_prepass_nt"): bit 0 = the pre-pass writes the rank tiles with nontemporal stores, bit 1 = reads the tuples with nontemporal loads
  // Ensembles with more than 32767 distinct thresholds on a feature (u16 ranks stop there; DTPU.sv:22,74 allows 8192 nodes x 64 PUs on one
  // feature) are scored in PARTS ("_cm" kernels only): consecutive chunks of the cluster-major image with rank tables of their own, one
  // pre-pass + scoring launch per part.  The reference-order sum runs THROUGH the parts: a launch starts from the accumulator and the
  // running total its predecessor left per tuple (state_in) after group0 PU groups, and leaves them (state_out) instead of the score.
  // feature compaction (ddt_choice.cpp): the kernels see tuples of ScoreArgs::tuple_words words, the caller's rows have in_words of them; the
  // pre-pass's transpose gathers column fmap[c] of a row into compact column c (~0: padding, reads as 0).  fmap == nullptr: no compaction
  const uint32_t* fmap = nullptr;
  uint32_t in_words = 0;
  const float* state_in = nullptr;   // [2][n_pad]: accumulator of the cluster in progress, running total over the finished clusters
  float* state_out = nullptr;
  uint32_t group0 = 0;               // PU groups (cluster-major order) in front of this launch's image
  // small batches on a "_cm_x" kernel (Variant::has_split): split > 1 = the launch is cut into `split` slices of the image -- grid (tiles, split),
  // ScoreArgs::out = a workspace of partial sums [positions][n_pad], added in the reference's order by launch_cm_combine behind the launch:
  //   split_len == 0: a slice = a cluster (split = min(clusters, PU groups that hold a real tree)), one partial per cluster (its accumulator)
  //   split_len  > 0: a slice = split_len consecutive chunks -- deep kernels: PU groups, two chunks each at CT = 4 -- (split = ceil(real chunks /
  //                   split_len)), one partial per PU group (its reduce tree); deep kernels: at position group0 + the group's place in this launch's image
  uint32_t split = 0, split_len = 0;
};
constexpr uint32_t kQ16TileCounterWords = 2;  // behind the kQ16GroupedCounters 8-byte counters

// ---------------------------------------------------------------------------------------------------
// Sparse (explicit-children) forests -- include/ddt.h ddt_load_model_sparse, kernel in ddt_sparse.hip.
//   top image   per PU group of 8 trees, per tree 12 * 2^K bytes: the first K levels as a PERFECT heap (early leaves
//               padded with dummy nodes): [0, 4*2^K) 8-byte records {thr_key, w} of levels 0..K-2 (1-based heap, record 0
//               padding), [4*2^K, 12*2^K) 2^(K-1) 16-byte records {thr_key, w, left, right} of level K-1
//   deep array  one 16-byte record {thr_key, w, left, right} per internal node at depth >= K, whole engine
//   w           = absolute LDS byte address of the feature row | kSpLeftLeaf | kSpRightLeaf | kSpMissRight: the two
//                 leaf flags sit in bits 31 / 30 so that each is ONE sign test (of w, of w << 1)
//   left/right  = index into the deep array (< 2^28: the kernel addresses it with a 32-bit byte offset), or the leaf's
//                 fp32 bits when the matching flag is set
//   dense level K ("sparse_dk_*", Variant::opt bit 1): the top image holds ALL K levels as 8-byte records (8 * 2^K bytes per
//               tree; record 0 = {cbase, 0}), and level K is a DENSE block of 2^K deep records per tree (a leaf or the padding
//               under an early leaf is a dummy record whose two children are that leaf value): the record below heap node m
//               of level K-1 on side s is deep[base + 2 (m - 2^(K-1)) + s], byte offset 2 * (8 * child heap index) + cbase with
//               cbase = 16 * base - 16 * 2^K (mod 2^32) -- no child words in LDS.  A third less LDS per tree: two K = 8 blocks of
//               256 tuples x 64 features share a CU (2 x 80 KiB), or one block of 512 holds K = 9 (160 KiB); the per-wave flags
//               of the missing-value test live at LDS offset 0 before the first image arrives.  EMPTY slots share the dummy
//               block deep[0, 2^K).
//   dense mid levels ("sparse_dm<M>_*", opt bit 3, Variant::top = M): the levels K .. K+M-1 continue the heap as 8-byte records {thr_key, w} in the
//               deep array (heap node h at byte cbase + 8 h, cbase = 16 * base - 8 * 2^K), the dense block of 16-byte records is level K+M
//               (byte cbase - 8 * 2^(K+M) + 16 h)
//   dense pair records ("sparse_dp_*", opt bit 4): the levels K and K+1 as ONE block of 2^K 16-byte records per tree {key of the level-K node,
//               key of its left child, key of its right child, feature NUMBER of the three in the bytes 0 / 1 / 2 + their missing directions
//               in the bits 24 / 25 / 26} at byte cbase + 16 h (cbase = 16 * base - 16 * 2^K, h = level-K heap index): one gather decides two
//               levels; the dense block of ordinary records is level K+2 (byte cbase - 32 * 2^K + 16 h).  Feature numbers < 256.
// ---------------------------------------------------------------------------------------------------
//   32-bit ranks, pair records on EVERY deep level ("sparse_r_*", Variant::opt bit 5, round 6; ddt_sparse_r.hip).  A node is ONE word
//               rec = R << 15 | feature number << 8 | kSrLeftLeaf | kSrRightLeaf | kSrMissRight  (feature number < 128),  R = 1 + index of the
//               threshold among the sorted distinct keys of its feature (< 2^17 - 1); the feature tile holds x' = rank(x) << 15 | 0x7FFF (a missing
//               value: 0xFFFFFFFF), written by the rank32 pre-pass, so that  !(x < t)  <=>  x' >= rec  as ONE unsigned compare, no mask.  The feature
//               number sits where the LDS address of its tile row has it (a row = 64 tuples x 4 bytes with "wave-private rows", Variant::wave_rows):
//               address = (rec & mask) | the lane's column, ONE VALU instruction (v_and_or_b32).
//               top image   per tree 4 * 2^K bytes: word 0 = cbase, words 1 .. 2^K - 1 the node words of levels 0..K-1 (1-based heap, early leaves
//                           padded with 0 = "feature 0 against rank 0": any direction ends on the same value)
//               deep array  16-byte PAIR records {node, left child, right child, ptr}: a child word is the child's node word, or the leaf's fp32 bits
//                           when the node carries the matching leaf flag; the record of the grandchild on side r1 of the internal child on side r0
//                           is at byte ptr + 32 r0 + 16 r1 (the host subtracts 32 from ptr when only the right child is internal: blocks of 2 or 4
//                           slots).  A grandchild that is a LEAF has a LEAF record {kSrLeafRec, value, value, 0} in its slot -- one more gather for that
//                           lane, no special case in the walk.  Level K is a dense block of 2^K pair records per tree: the record of heap node h (2^K <= h < 2^(K+1))
//                           is at byte cbase + 16 h, cbase = the block's byte offset - 16 * 2^K (mod 2^32; the kernel carries 4 h).  EMPTY slots share a block of LEAF(+0).
//               One gather decides two levels everywhere below the top image: 512 x depth 16 with K = 9: 4 gather instructions per tree and wave (7 before).
constexpr uint32_t kSpLeftLeaf = 0x80000000u, kSpRightLeaf = 0x40000000u, kSpMissRight = 0x20000000u, kSpAddrMask = 0x1FFFFFFFu;
constexpr int kSparseMinTop = 6, kSparseMaxTop = 10;
constexpr uint32_t kSrLeftLeaf = 0x80u, kSrRightLeaf = 0x40u, kSrMissRight = 0x20u, kSrFeatShift = 8u, kSrFeatMask = 0x7F00u, kSrRankShift = 15u;
constexpr uint32_t kSrLeafRec = kSrLeftLeaf | kSrRightLeaf;  // node word of a LEAF record: rank 0, feature 0, both sides leaves
constexpr uint32_t kSrMissing = 0xFFFFFFFFu;                 // a missing value in the 32-bit rank tile
constexpr uint32_t kSrMaxTable = (1u << 17) - 2u;            // distinct thresholds per feature (17-bit ranks)
constexpr uint32_t kSrMaxWords = 128;                        // tuple words (the pre-pass's transpose stages 256 rows x (W + 1) words in LDS)
constexpr uint32_t kSrMaxDir = 32767;                        // directory entries per feature (rank32_kernel keeps one feature's directory in LDS)

// 32-bit rank pre-pass (ddt_sparse_r.hip rank32_kernel).  Per feature the sorted distinct keys in BLOCKS of 2^blk_log2 keys (>= 4; padded with
// INT_MAX, one all-pad block behind them) in `tab`, and a DIRECTORY = the last key of every block, searched out of LDS exactly like rank_kernel's
// table (Q16Aux::tables / tabP / tabS / Kpad hold the directory); the block the directory names is then read from `tab` with one 16-byte gather
// per four keys.  tabP[j] = {directory entries, lo, hi, shift, P, key offset of the feature's blocks in tab, K (real keys), largest key}.
struct R32Aux {
  uint32_t* r = nullptr;           // workspace [n_pad / T][W][T] u32, T = the scoring kernel's tile
  const uint32_t* tab = nullptr;
  uint32_t tab_bytes = 0;
  uint32_t blk_log2 = 2;
  uint32_t tile = 256;
};

struct SparseAux {           // ScoreArgs::aux of the sparse kernels
  const uint4* deep;         // deep records (at least one, record 0 is a valid dummy)
  uint32_t n_groups;         // PU groups of 8 trees in the top image
  uint32_t deep_bytes;       // bytes of the deep array: the range of the buffer resource its records are gathered through
  uint32_t max_rounds;       // visits of the deep loop after which every walker is at a leaf (>= 1; from the forest's deepest path and the kernel's K / M):
                             // the last of them issues no gather (round 5)
  uint32_t idle_off;         // byte offset a FINISHED lane's (unconditional) gather uses: 0 = record 0 (rounds 2-4), 0xFFFFFFF0 = beyond the
                             // resource's range -- the lane gets 0 back and no cache access is made (round 5, option "sparse_idle_oob")
  // rank-quantised sparse kernels ("sparse_q_*", Variant::opt bit 0): thresholds are ranks, the features arrive as the u16 tiles
  // of the q16 pre-pass (the same tables / workspace / kernels as the perfect-tree q16 path); slow images = the same images
  Q16Aux q16;
  R32Aux r32;                // "sparse_r_*": the 32-bit rank pre-pass (q16.xT / tile_flags / tables / tabP / tabS / Kpad / n_pad / skip_prepass are shared)
};

enum { kKindGeneric = 0, kKindTile = 1, kKindStream = 2, kKindQ16 = 3, kKindSparse = 4 };

constexpr uint32_t kStreamSkew = 64u;  // bytes, see Variant::feat_word_stream
struct Variant {
  const char* name;
  int kind;
  int levels;         // compile-time D, 0 = generic (runtime D)
  int threads;        // block size
  int tuples_per_lane;
  int chunk_trees;    // trees per LDS chunk (tile kernels); padding granule otherwise
  int ilp_trees;      // trees walked concurrently per lane
  int stage;          // tile: 0 = model chunks staged through registers, 1 = global->LDS DMA
  int opt;            // tile: bit 0 = last level fused with its leaves (image layout 1); stream: max lines per tuple
  hipError_t (*launch)(const ScoreArgs&, const Variant&, hipStream_t);
  int top = 0;        // rank-quantised DEEP kernels ("q16d_*", opt bit 5): K, the levels of a tree staged in LDS; the levels below are gathered

  uint32_t tile() const { return (uint32_t)threads * (uint32_t)tuples_per_lane; }
  uint32_t row_bytes() const { return tile() * 4u; }
  uint32_t tree_bytes() const { return 12u << levels; }
  uint32_t chunk_bytes() const { return tree_bytes() * (uint32_t)chunk_trees; }
  // ---- tile kernels: LDS = [model_base, +2 chunks) double buffer, then the feature tile ----
  // layout 1 addresses the last level as 2*m8 + (tree base - 4*2^D): the model buffers start 4*2^D bytes
  // into LDS so that this DS immediate is never negative
  uint32_t model_base() const { return (opt & 1) ? (4u << levels) : 0u; }
  uint32_t feat_off() const {  // multiple of the row size so that (row offset | lane offset) is an OR
    const uint32_t row = row_bytes(), need = model_base() + 2u * chunk_bytes();
    return (need + row - 1u) / row * row;
  }
  uint32_t lds_bytes(uint32_t tuple_words) const { return feat_off() + tuple_words * row_bytes() + 64u; }  // +64: per-wave flags
  // ---- stream kernels: LDS = [0, image) resident model, then the feature tile ----
  uint32_t feat_off_stream(uint32_t n_trees_padded) const {
    const uint32_t row = row_bytes(), need = n_trees_padded * tree_bytes();
    return (need + row - 1u) / row * row;
  }
  // the rows of tuple line q (features 4q..4q+3) start kStreamSkew * q bytes late: the four lanes that stage the four lines
  // of one tuple then hit four different groups of 16 banks (conflict-free transposed stores; it was 4-way)
  uint32_t feat_word_stream(uint32_t n_trees_padded, uint32_t j) const {
    return feat_off_stream(n_trees_padded) + j * row_bytes() + kStreamSkew * (j / 4u);
  }
  uint32_t lds_bytes_stream(uint32_t n_trees_padded, uint32_t tuple_words) const {
    return feat_off_stream(n_trees_padded) + tuple_words * row_bytes() + kStreamSkew * (tuple_words / 4u) + 64u;
  }
  // ---- q16 kernels: 4-byte node records + fp32 leaves = 8*2^D bytes per tree; u16 feature tile ----
  // opt bit 0 ("_gl"): only the node records are staged in LDS (4*2^D bytes per tree); the leaves stay in global memory and
  // are gathered through the vector-memory path.  The image keeps 8*2^D bytes per tree either way: per chunk the records
  // of its trees, then (opt 1) the leaves of its trees / (opt 0) records and leaves tree by tree.
  // ---- deep rank-quantised kernels ("q16d_dD_kK_*", opt bit 5; always cluster-major, opt bit 2): perfect trees of depth D > K.  Per tree
  //   top     4 * 2^K bytes: the records of levels 0..K-1 (1-based heap, record 0 padding) -- the part of a chunk that is staged in LDS
  //   stages  G = (D - K + 1) / 2 blocks of 16-byte records (D - K is odd), gathered through a buffer resource, ONE gather per stage:
  //           stage g < G-1 ("pair", level L = K + 2g, 2^L records): {record of node i, record of its left child, of its right child, 64 * i}
  //                         -- two levels per gather; the next stage's record of the grandchild is at byte  w + 32 * right0 + 16 * right1
  //           stage G-1     ("terminal", level D-1, 2^(D-1) records): {record of node i, left leaf, right leaf, 0}
  //   Records are the q16 records {rank (lo16) | feature row byte offset (hi16; bit 16 = miss_right in the slow image)}.
  //   Image = per chunk of chunk_trees trees their tops, then their stage blocks tree by tree.
  bool deep() const { return kind == kKindQ16 && (opt & 32) != 0; }
  // opt bit 6 ("q16w_*" / "q16dw_*"): tuples of up to 64 words -- a record's row-offset field holds HALF the byte offset (the kernel shifts it
  // back), the block takes up to 144 KiB of LDS (one block of 16 waves per CU)
  bool wide() const { return kind == kKindQ16 && (opt & 64) != 0; }
  // the plain rank-quantised kernels the automatic choice takes ("q16_d8_c8_u4_gl_s2_cm_x", its wide form, "q16_d{5,6,7}_*_s2", "q16_d{3,4}_*")
  // have a second instantiation whose grid is cut into slices of the image (Q16Aux::split): what launch_score gives a batch of a few tiles
  bool has_split() const {
    if (kind != kKindQ16 || (opt & 8) != 0) return false;                                          // (not the persistent "_p" form)
    if (opt & 32) return true;                                                                      // the deep kernels: all of them (launch_q16d)
    if (chunk_trees % 8 != 0) return false;
    return levels == 8 ? (opt & 4) != 0 : levels >= 5 ? (opt & 2) != 0 : levels >= 3;               // = the instantiations of launch_q16 (ddt_kernels.hip)
  }
  bool cm() const { return kind == kKindQ16 && (opt & 4) != 0; }                                    // cluster-major image
  uint32_t max_tuple_words_q16() const { return wide() ? 64u : 32u; }
  uint32_t deep_stages() const { return ((uint32_t)levels - (uint32_t)top + 1u) / 2u; }
  uint32_t deep_stage_level(uint32_t g) const { return g + 1u < deep_stages() ? (uint32_t)top + 2u * g : (uint32_t)levels - 1u; }
  uint32_t deep_stage_off(uint32_t g) const {  // byte offset of stage g inside a tree's stage blocks
    uint32_t off = 0;
    for (uint32_t k = 0; k < g; ++k) off += 16u << deep_stage_level(k);
    return off;
  }
  uint32_t deep_bytes() const { return deep_stage_off(deep_stages()); }
  uint32_t tree_bytes_q16() const { return deep() ? (4u << top) + deep_bytes() : (8u << levels); }
  uint32_t lds_tree_bytes_q16() const { return deep() ? (4u << top) : (opt & 1) ? (4u << levels) : (8u << levels); }
  uint32_t feat_off_q16() const { return 2u * lds_tree_bytes_q16() * (uint32_t)chunk_trees; }
  // opt bit 1 ("_s2"): the records of levels 0-1 come from SGPRs (scalar loads), see ddt_kernels.hip
  uint32_t lds_bytes_q16(uint32_t tuple_words) const { return feat_off_q16() + tuple_words * tile() * 2u; }
  // ---- sparse kernels (levels = K, the top levels staged in LDS): LDS = [top image of one PU group][feature tile] ----
  // opt bit 0 ("sparse_q_*"): rank-quantised -- u16 feature tile (half the LDS per tuple: 1024 tuples = 16 waves share a CU
  // where the fp32 tile holds 512), node thresholds are ranks
  // opt bit 1 ("sparse_dk_*"): dense level K -- 8-byte records only in LDS, no flag bytes behind the tile (above: "Sparse forests")
  // opt bit 5 ("sparse_r_*"): 32-bit ranks -- one-word nodes in the top image, a u32 feature tile written by the rank32 pre-pass, pair records below
  bool r32() const { return kind == kKindSparse && (opt & 32) != 0; }
  uint32_t top_bytes_sparse() const { return ((opt & 32) ? 4u : (opt & 2) ? 8u : 12u) << levels; }
  // opt bit 2 ("sparse_gf_*", any tuple width): no feature tile -- a feature is gathered from the tuple's row in global memory, the
  // record's address field is the byte offset of the feature inside the row (row 0 at 0, 4 bytes per feature)
  uint32_t row_bytes_sparse() const { return (opt & 4) ? 4u : (opt & 1) ? tile() * 2u : tile() * 4u; }
  uint32_t feat_off_sparse() const {  // chunk_trees = trees walked in lock-step = top images resident per pass
    if (opt & 4) return 0u;
    const uint32_t row = row_bytes_sparse(), need = (uint32_t)chunk_trees * top_bytes_sparse();
    return (need + row - 1u) / row * row;
  }
  // "sparse_r_*": WAVE-PRIVATE ROWS -- the tile in LDS as [wave][feature][64 tuples], a wave's region padded to a power of two and aligned to it,
  // so that a row's address is (node word & mask) | lane column, one VALU instruction (ddt_sparse_r.hip SrTile).  Taken for tuples of up to 64 words
  // when the padding costs no block per CU (20 words pad to 32: three blocks of K = 10 become two -- measured 1361 vs 1600 Mtuples/s on 128 x d14 x 20)
  static uint32_t wave_region_bytes(uint32_t tuple_words) {
    uint32_t ws = 1024u;  // (tuple_words is a multiple of 4: whole 1 KiB DMA units)
    while (ws < tuple_words * 256u) ws <<= 1;
    return ws;
  }
  uint32_t lds_bytes_sparse_rows(uint32_t tuple_words, bool wave_rows) const {
    const uint32_t need = (uint32_t)chunk_trees * top_bytes_sparse();
    if (wave_rows) {
      const uint32_t ws = wave_region_bytes(tuple_words);
      return (need > ws ? need : ws) + (tile() / 64u) * ws;
    }
    return feat_off_sparse() + tuple_words * row_bytes_sparse();
  }
  bool wave_rows(uint32_t tuple_words) const {
    if (!r32() || tuple_words > 64u) return false;
    constexpr uint32_t lds_per_cu = 160u * 1024u;  // MI355X
    return lds_per_cu / lds_bytes_sparse_rows(tuple_words, true) >= lds_per_cu / lds_bytes_sparse_rows(tuple_words, false);
  }
  uint32_t lds_bytes_sparse(uint32_t tuple_words) const {
    if (r32()) return lds_bytes_sparse_rows(tuple_words, wave_rows(tuple_words));
    if (opt & 4) return (uint32_t)chunk_trees * top_bytes_sparse() + 64u;
    return feat_off_sparse() + tuple_words * row_bytes_sparse() + ((opt & (2 | 32)) ? 0u : 64u);
  }
};

int num_variants();
const Variant& variant(int i);

// generic kernel geometry (runtime D; see ddt_kernels.hip)
constexpr int kGenericThreads = 256;
hipError_t launch_generic(const ScoreArgs& a, const Variant& v, hipStream_t s);
uint32_t generic_lds_bytes(uint32_t levels, uint32_t tuple_words, bool* feat_in_lds, bool* tree_in_lds, uint32_t* top_levels);
uint32_t stream_blocks_per_cu(uint32_t lds_bytes);

hipError_t launch_q16_prepass(const ScoreArgs& a, const Q16Aux& x, hipStream_t s);  // ddt_prepass.hip: rank pre-pass of one batch

constexpr int kQTile = 1024;               // tuples per u16 rank tile == threads of a rank-quantised scoring block
int num_deep_variants();                   // ddt_deep.hip: deep perfect trees, appended to the variant table after the kernels of ddt_kernels.hip
const Variant& deep_variant(int i);
int num_sparse_variants();                 // ddt_sparse.hip: appended after those
const Variant& sparse_variant(int i);
int num_sparse_r_variants();               // ddt_sparse_r.hip: and these last
const Variant& sparse_r_variant(int i);
hipError_t launch_transpose(const uint32_t* tuples, uint32_t W, uint64_t n, uint64_t n_pad, uint32_t* xT, hipStream_t s);  // ddt_prepass.hip
// zeroes `words` 32-bit words with a KERNEL (tile flags, ticket counters): what every call clears in front of its pre-pass.  Not hipMemsetAsync: inside a
// graph that PyTorch captured and replays, the memset node in front of the ticket-counter kernels did not reliably clear the counters (replays ranked
// nothing and scored the previous batch's ranks; the same calls captured with plain hipStreamBeginCapture were fine) -- a kernel node has no such mode
hipError_t launch_zero_words(uint32_t* p, uint64_t words, hipStream_t s);  // ddt_prepass.hip

// n_classes > 1: the classes of a one-vs-all model, class_positions partial sums each, class k's sum to out[k * out_pitch + row]
hipError_t launch_cm_combine(const float* parts, size_t pitch, size_t n, uint32_t real_groups, uint32_t clusters, bool per_group, bool cm_order, float* out,
                             bool exact /* sum_mode 2 */, hipStream_t s, uint32_t n_classes = 1, uint32_t class_positions = 0, size_t out_pitch = 0);
hipError_t launch_chain_sum(const float* parts, uint32_t n_parts, size_t n, float* out, bool exact /* sum_mode 2 */, hipStream_t s,
                            size_t pitch = 0 /* elements between the partial vectors; 0 = n */);
hipError_t launch_argmax(const float* scores, uint32_t K, size_t n, int32_t* labels, hipStream_t s);
hipError_t launch_argmax_strided(const float* scores, uint32_t K, size_t pitch, size_t n, int32_t* labels, hipStream_t s);
hipError_t launch_synth_tuples(uint32_t* out, uint64_t row0, size_t n, uint32_t F, int dist, uint32_t missing_bits,
                               hipStream_t s);

// The reference's fp32 adder on the exact leaf domain (sum_mode 2; the one case where it is not the IEEE add is described in
// ddt_device.h).  Host + device: the kernels call it on their rare path, the CPU model of the kernels (tests/mock_hip) always.
__host__ __device__ inline uint32_t f32_bits(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}
__host__ __device__ inline float ref_add_exact(float a, float b) {
  const float s = a + b;
  const uint32_t ua = f32_bits(a), ub = f32_bits(b);
  const uint32_t ma = ua & 0x7FFFFFFFu, mb = ub & 0x7FFFFFFFu;
  const uint32_t big = ma >= mb ? ua : ub, small = ma >= mb ? ub : ua;
  const uint32_t eb = (big >> 23) & 0xFFu, es = (small >> 23) & 0xFFu;
  // FPAdder_2cycles_latency.v:325-326: effective subtraction, larger operand a power of two, exponents exactly 25 apart,
  // smaller mantissa != 0 -> the RTL returns the larger operand, IEEE the float just below it
  const bool corner = (big & 0x7FFFFFu) == 0u && ((ua ^ ub) >> 31) != 0u && eb - es == 25u && es != 0u && (small & 0x7FFFFFu) != 0u;
  float r;
  __builtin_memcpy(&r, &big, 4);
  return corner ? r : s;
}

// splitmix64 / synthetic definitions; shared by host generator and kernels
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
constexpr uint64_t kSeedX = 0x0DD7000000000001ull;
constexpr uint64_t kSeedM = 0x0DD7000000000002ull;
constexpr uint64_t kSeedS = 0x0DD7000000000003ull;  // synthetic sparse forests (ddt_synth_sparse_model)

// IEEE order-preserving key (cmp_mode 1): signed-int compare of keys == IEEE '<' on the floats;
// every NaN -> INT_MAX (never "less"), -0 -> +0.
__host__ __device__ inline uint32_t ieee_key(uint32_t b) {
  if ((b & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FFFFFFFu;
  if (b == 0x80000000u) return 0u;
  return (b & 0x80000000u) ? (b ^ 0x7FFFFFFFu) : b;
}

}  // namespace ddt
