// pcv_spec.h — host logic of the single-chain ("speculative") build.
//
// The exact build runs the per-point quantise->decode chain twice: once for the path keys that give the topology
// (K2 + key sort + node split) and once more to the leaf each point ends in (K5). The chain is f64-VALU bound and is
// 44 % of the build. The single-chain build predicts the topology from a strided SAMPLE, walks every point down the
// predicted tree in ONE chain pass, counts the points per predicted leaf exactly, and derives the TRUE tree from those
// exact counts. Nothing about the result is approximate:
//   * the predicted tree T'' only has to be at least as deep as the true tree wherever points go; every inner node has
//     all eight children, so every point reaches exactly one predicted leaf;
//   * a node whose sampled count is close to the capacity ("candidate") is split in T''; a point passing through one
//     leaves the chain pass with the codes it had AT the FIRST candidate on its path (the walk still goes on to the
//     predicted leaf, whose exact count the split decisions need). The chain of a point from level k on is a function
//     of its level-k codes alone (every level re-decodes the position from the codes, generation.rs:78-99), so those codes
//     serve both outcomes: if the candidate is a true leaf they ARE the leaf codes — nothing to patch —, and if it is
//     split the chain is CONTINUED from them, leaf by leaf, once the record sort has made each true leaf contiguous;
//   * the exact counts decide (should_split_node, reference src/octree/generation.rs:128-150). A true leaf that is a
//     non-candidate inner node of T'' with no candidate above it (a count far outside the band: rare) has no usable
//     codes — they may belong to a level BELOW it — and its points replay the chain from their coordinates; a
//     predicted leaf that must be split (the prediction is too shallow there) sends the whole build to the exact
//     two-chain pipeline. Speculation can cost time, never correctness — the same contract as the depth speculation of
//     the exact path.
#pragma once
#include <stdint.h>
#include <stdlib.h>

// Experiment switches (environment variables, each read once per process) exist only in libpcv_hip_exp.so, the build made
// with -DPCV_EXPERIMENTS for the A/B scripts under tools/ and for the tests of the alternative kernels. The shipped
// library reads no environment variable.
inline const char* pcv_experiment(const char* name) {
#ifdef PCV_EXPERIMENTS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

#include <vector>

// Walk record of a T'' node (device, 32 bits so that the table of a 100 M-point tree fits the 32 KiB vector L1): inner:
// first child index in bits 0..29 (the eight children are consecutive), leaf: predicted-leaf rank in bits 0..29.
#define PCV_SPEC_LEAF (1u << 31)
#define PCV_SPEC_CANDIDATE (1u << 30)
#define PCV_SPEC_INDEX_MASK 0x3fffffffu
// predicted-leaf -> true-leaf map entry: true rank in bits 0..29, bit 30: no usable codes (the point leaves its input
// index in the payload and replays the chain from its coordinates after the record sort)
#define PCV_SPEC_MAP_REPLAY (1u << 30)

struct PcvSpecParams {
  uint32_t cap = 0;         // max_points_per_node
  double resolution = 0;
  const double* edge = nullptr;  // edge[k], k = 0 .. nlevels
  int nlevels = 0;          // digit levels of the sample keys (<= 21)
  uint32_t force_mask = 0;  // level-1 nodes that are split whatever they hold (multi-GPU build)
  double scale = 1;         // N / S: points per sample point
  double delta = 0;         // relative half width of the candidate band around the capacity
};

// Node table of the SAMPLE tree as pcv_launch_node_split leaves it (BFS order, children contiguous in digit order).
struct PcvSampleTable {
  uint32_t num_nodes = 0;
  const uint64_t* prefix = nullptr;
  const uint32_t* lo = nullptr;
  const uint32_t* hi = nullptr;
  const uint32_t* first_child = nullptr;
  const uint8_t* level = nullptr;
  const uint8_t* child_mask = nullptr;
  const uint8_t* open = nullptr;
};

struct PcvSpecTree {
  std::vector<uint64_t> prefix;       // left-aligned path key, 3 bits per level, level 1 at bits 60..62
  std::vector<uint8_t> level;
  std::vector<uint8_t> inner;
  std::vector<uint8_t> candidate;     // inner node whose sampled count is inside the band
  std::vector<uint32_t> first_child;  // inner nodes: index of child 0 (children 0..7 follow each other)
  std::vector<uint32_t> parent;
  std::vector<uint32_t> leaf_rank;    // the name a predicted leaf writes into the rank array: its node index
  std::vector<uint32_t> walk;         // device walk records, one per node
  uint32_t num_leaves = 0;            // number of counter bins (== number of nodes)
  bool any_candidate = false;
};
// Host view of a predicted tree built on the device (walk records + parent + level per node); false: inconsistent.
bool pcv_spec_tree_from_walk(const uint32_t* walk, const uint32_t* parent, const uint8_t* level, uint32_t count, PcvSpecTree* out);

// Sample split threshold: a sample node is opened iff its sample count exceeds this (count * scale > cap * (1 - delta)).
uint32_t pcv_spec_sample_threshold(const PcvSpecParams& p);

void pcv_spec_build_tree(const PcvSpecParams& p, const PcvSampleTable& s, PcvSpecTree* out);

enum PcvSpecStatus {
  PCV_SPEC_OK = 0,
  PCV_SPEC_TOO_SHALLOW = 1,  // a predicted leaf holds more than the capacity and may be split: redo with the exact pipeline
};

// The true tree in the layout the exact path downloads from the device after the node split (BFS order, children
// contiguous in digit order, [lo, hi) = range in key-sorted order), plus the map predicted-leaf rank -> true leaf rank
// (depth-first order, the same order pcv_build_finish assigns) with PCV_SPEC_MAP_REPLAY set when the points have to
// replay their chain from the coordinates.
struct PcvTrueTree {
  std::vector<uint64_t> prefix;
  std::vector<uint32_t> lo, hi, first_child;
  std::vector<uint8_t> level, child_mask, open;
  std::vector<uint32_t> spec_map;
  // A true leaf that is a non-candidate inner node of T'' without a candidate above it: level of that leaf per predicted
  // leaf below it, 0 = none. Those points replay the chain to that level once the record sort has made them contiguous
  // (rare; any_fix says whether the table is needed).
  std::vector<uint8_t> fix_level;
  std::vector<uint32_t> fix_nodes;  // the true leaves (indices into this table) whose points replay the chain
  bool any_fix = false;
  // True leaves BELOW the first candidate of their path (the candidate was split): their records carry the codes of the
  // candidate's level. cont_nodes[j] = the leaf, cont_from[j] = that candidate (both indices into this table); the chain
  // is continued from the one level to the other over the leaf's sorted slots.
  std::vector<uint32_t> cont_nodes, cont_from;
  uint64_t fix_points = 0, kept_points = 0, cont_points = 0;  // replayed / in an unsplit first candidate / continued
  uint32_t num_leaves = 0;
  int deepest_level = 0;
};
PcvSpecStatus pcv_spec_resolve(const PcvSpecParams& p, const PcvSpecTree& t, const uint32_t* leaf_counts, PcvTrueTree* out);

// ---- K6 work lists (host) ----------------------------------------------------------------------------------------------
// One workgroup of the leaf-wise `settle` / `climb` kernels: up to kPcvSettleTile consecutive sorted slots — resp. up to
// kPcvClimbTile consecutive climber records — [begin, end) of ONE leaf. The host knows every leaf's slot range when it
// builds the node tables, so a workgroup gets its leaf's record through scalar loads that run beside its record loads
// instead of behind them.
struct alignas(16) PcvSettleItem {
  uint32_t rank, begin, end, pad;
};
// Four slots per lane: `settle` lives on the record loads it has in flight (r03t A/B at 100 M points, one box: 512 slots per
// workgroup 0.680-0.683 ms, 1 024 0.637-0.639, 2 048 0.663 — 124 VGPRs, four waves per SIMD)
#ifndef PCV_SETTLE_TILE
#define PCV_SETTLE_TILE 1024
#endif
constexpr uint32_t kPcvSettleTile = PCV_SETTLE_TILE;
constexpr uint32_t kPcvClimbTile = 256;
// Leaves in rank order: leaf r holds the sorted slots [lo[r], lo[r] + count[r]). Writes the settle items (in slot order)
// and returns their number (<= n / kPcvSettleTile + num_leaves).
uint32_t pcv_settle_items(const uint32_t* lo, const uint32_t* count, uint32_t num_leaves, PcvSettleItem* out);
// Climbers: every 8th point (j % 8 == 0) of a leaf whose node is not the root (climbs[r] != 0) — ceil(count / 8) records,
// dense from climb_base[r] (filled here: exclusive prefix sum in rank order). Writes the climb items, stores their number
// in *num_items (<= total / kPcvClimbTile + num_leaves) and returns the total number of climber records. A climb item's
// pad is its leaf's climb_base (with 16-byte climber records a record's index is all that says which slot it came from).
uint64_t pcv_climb_layout(const uint32_t* count, const uint8_t* climbs, uint32_t num_leaves, uint32_t* climb_base,
                          PcvSettleItem* out, uint32_t* num_items);
