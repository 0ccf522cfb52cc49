// pcv_spec.cpp — host logic of the single-chain build (see pcv_spec.h). Plain C++, no HIP: unit-tested on the CPU
// through pcv_spec_selftest (tests/test_spec_cpu.py) against the oracle's tree.
#include "pcv_spec.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "../../include/pcv_hip.h"

namespace {
constexpr int kKeyLevels = PCV_MAX_KEY_LEVELS;  // 21 levels x 3 bits, level 1 at the top of the 63-bit key

inline uint64_t digit_bits(unsigned c, int level) { return (uint64_t)c << (3 * (kKeyLevels - level)); }
}  // namespace

uint32_t pcv_spec_sample_threshold(const PcvSpecParams& p) {
  // open a sample node iff count_s * scale > cap * (1 - delta)  <=>  count_s > floor(cap * (1 - delta) / scale)
  const double t = std::floor((double)p.cap * (1.0 - p.delta) / p.scale);
  return t <= 0.0 ? 0u : (t >= 4294967294.0 ? 4294967294u : (uint32_t)t);
}

void pcv_spec_build_tree(const PcvSpecParams& p, const PcvSampleTable& s, PcvSpecTree* out) {
  PcvSpecTree& t = *out;
  t = PcvSpecTree();
  // a sample node inside the band: opened (count_s > lower threshold) but count_s * scale <= cap * (1 + delta)
  const double upper = (double)p.cap * (1.0 + p.delta) / p.scale;
  std::vector<int64_t> sample_of;  // T'' node -> sample node, -1 = no sample point fell in it
  auto push = [&](uint64_t prefix, int level, uint32_t parent, int64_t sample_node) {
    t.prefix.push_back(prefix);
    t.level.push_back((uint8_t)level);
    t.inner.push_back(0);
    t.candidate.push_back(0);
    t.first_child.push_back(0);
    t.parent.push_back(parent);
    t.leaf_rank.push_back(0);
    sample_of.push_back(sample_node);
  };
  push(0, 0, 0xffffffffu, s.num_nodes ? 0 : -1);
  for (size_t i = 0; i < t.prefix.size(); ++i) {  // breadth first: the vectors grow while we walk them
    const int64_t sn = sample_of[i];
    if (sn < 0 || !s.open[sn]) continue;
    const int level = t.level[i];
    t.inner[i] = 1;
    t.first_child[i] = (uint32_t)t.prefix.size();
    const double cnt = (double)(s.hi[sn] - s.lo[sn]);
    const bool forced = level == 0 || (level == 1 && ((p.force_mask >> ((t.prefix[i] >> (3 * (kKeyLevels - 1))) & 7)) & 1u));
    if (!forced && cnt <= upper) {
      t.candidate[i] = 1;
      t.any_candidate = true;
    }
    uint32_t next = s.first_child[sn];
    const uint64_t pfx = t.prefix[i];
    for (unsigned c = 0; c < 8; ++c) {
      const bool present = (s.child_mask[sn] >> c) & 1;
      push(pfx | digit_bits(c, level + 1), level + 1, (uint32_t)i, present ? (int64_t)next : -1);
      if (present) ++next;
    }
  }
  // a predicted leaf is named by its node index (the rank map takes any unique name); every node gets a counter bin
  t.num_leaves = (uint32_t)t.prefix.size();
  t.walk.resize(t.prefix.size());
  for (size_t i = 0; i < t.prefix.size(); ++i) {
    t.leaf_rank[i] = (uint32_t)i;
    t.walk[i] = t.inner[i] ? (t.first_child[i] | (t.candidate[i] ? PCV_SPEC_CANDIDATE : 0u)) : ((uint32_t)i | PCV_SPEC_LEAF);
  }
}

// The host's view of a predicted tree that was built on the DEVICE (spec_tree_emit_kernel): walk records, parent and
// level per node. Children follow their parent (the k-th inner node's children sit at 1 + 8 k .. 1 + 8 k + 7).
bool pcv_spec_tree_from_walk(const uint32_t* walk, const uint32_t* parent, const uint8_t* level, uint32_t count, PcvSpecTree* out) {
  PcvSpecTree& t = *out;
  t = PcvSpecTree();
  t.prefix.assign(count, 0);
  t.level.assign(level, level + count);
  t.inner.assign(count, 0);
  t.candidate.assign(count, 0);
  t.first_child.assign(count, 0);
  t.parent.assign(parent, parent + count);
  t.leaf_rank.resize(count);
  t.walk.assign(walk, walk + count);
  t.num_leaves = count;
  for (uint32_t i = 0; i < count; ++i) {
    t.leaf_rank[i] = i;
    const uint32_t rec = walk[i];
    if (!(rec & PCV_SPEC_LEAF)) {
      const uint32_t first = rec & PCV_SPEC_INDEX_MASK;
      if (first <= i || (uint64_t)first + 8 > count) return false;  // children must follow their parent and exist
      t.inner[i] = 1;
      t.first_child[i] = first;
      if (rec & PCV_SPEC_CANDIDATE) {
        t.candidate[i] = 1;
        t.any_candidate = true;
      }
    } else if ((rec & PCV_SPEC_INDEX_MASK) != i) {
      return false;
    }
    if (i) {
      const uint32_t p = parent[i];
      if (p >= i || !t.inner[p] || i < t.first_child[p] || i >= t.first_child[p] + 8 || level[i] != level[p] + 1 ||
          level[i] > kKeyLevels)
        return false;
      t.prefix[i] = t.prefix[p] | digit_bits(i - t.first_child[p], level[i]);
    } else if (level[0] != 0) {
      return false;
    }
  }
  return true;
}

PcvSpecStatus pcv_spec_resolve(const PcvSpecParams& p, const PcvSpecTree& t, const uint32_t* leaf_counts, PcvTrueTree* out) {
  // This runs with the GPU idle (exact counts in, rank map out): flat pre-sized arrays, no allocation inside the loops.
  PcvTrueTree& r = *out;
  r = PcvTrueTree();
  const size_t m = t.prefix.size();
  // exact point count of every T'' node: children follow their parent in the (breadth-first) table
  std::vector<uint64_t> cnt(m);
  for (size_t i = m; i-- > 0;) {
    if (!t.inner[i]) {
      cnt[i] = leaf_counts[t.leaf_rank[i]];
    } else {
      const uint64_t* c = cnt.data() + t.first_child[i];
      cnt[i] = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
    }
  }
  r.spec_map.assign(t.num_leaves, 0);
  r.fix_level.assign(t.num_leaves, 0);
  if (m == 0 || cnt[0] == 0) return PCV_SPEC_OK;  // no points: no nodes (the caller handles n == 0 before it gets here)

  // should_split_node (generation.rs:128-150) with the EXACT counts; the root is always split (generation.rs:312-323)
  auto split = [&](uint32_t i) {
    const int level = t.level[i];
    if (level == 0) return true;
    if (level == 1 && ((p.force_mask >> ((t.prefix[i] >> (3 * (kKeyLevels - 1))) & 7)) & 1u)) return true;
    return cnt[i] > (uint64_t)p.cap && p.edge[level] > p.resolution;
  };
  // true tree, breadth first (a subtree of T'': at most m nodes); src[k] = T'' node of true node k
  std::vector<uint32_t> src(m);
  // first candidate node on the path from the root to (but excluding) this node, as an index into the true tree;
  // kNone: no candidate above. The chain pass left every point below it with the codes of THAT node's level.
  constexpr uint32_t kNone = 0xffffffffu;
  std::vector<uint32_t> first_cand(m);
  r.prefix.resize(m);
  r.level.resize(m);
  r.lo.resize(m);
  r.hi.resize(m);
  r.first_child.resize(m);
  r.child_mask.resize(m);
  r.open.resize(m);
  size_t nk = 1;
  src[0] = 0;
  first_cand[0] = kNone;
  for (size_t k = 0; k < nk; ++k) {
    const uint32_t i = src[k];
    r.prefix[k] = t.prefix[i];
    r.level[k] = t.level[i];
    if (t.level[i] > r.deepest_level) r.deepest_level = t.level[i];
    if (!split(i)) continue;
    if (!t.inner[i]) return PCV_SPEC_TOO_SHALLOW;  // the prediction stops above where the tree goes on
    r.open[k] = 1;
    r.first_child[k] = (uint32_t)nk;
    uint8_t mask = 0;
    const uint32_t fc = first_cand[k] != kNone ? first_cand[k] : (t.candidate[i] ? (uint32_t)k : kNone);
    for (unsigned c = 0; c < 8; ++c) {
      const uint32_t ch = t.first_child[i] + c;
      if (cnt[ch] == 0) continue;  // a child exists iff a point lies in it
      mask |= (uint8_t)(1u << c);
      src[nk] = ch;
      first_cand[nk] = fc;
      ++nk;
    }
    r.child_mask[k] = mask;
  }
  r.prefix.resize(nk);
  r.level.resize(nk);
  r.lo.resize(nk);
  r.hi.resize(nk);
  r.first_child.resize(nk);
  r.child_mask.resize(nk);
  r.open.resize(nk);
  // leaves in key order (depth first, digits ascending — the order pcv_build_finish ranks them in): ranges in the
  // sorted order and the predicted-leaf -> true-leaf map
  std::vector<uint32_t> stack(8 * (size_t)(kKeyLevels + 2)), below(8 * (size_t)(kKeyLevels + 2));
  size_t sp = 0;
  stack[sp++] = 0;
  uint64_t run = 0;
  while (sp) {
    const uint32_t k = stack[--sp];
    if (r.open[k]) {
      const uint32_t nchild = (uint32_t)__builtin_popcount(r.child_mask[k]);
      for (uint32_t c = nchild; c-- > 0;) stack[sp++] = r.first_child[k] + c;
      continue;
    }
    const uint32_t i = src[k];
    const uint32_t rank = r.num_leaves++;
    r.lo[k] = (uint32_t)run;
    run += cnt[i];
    r.hi[k] = (uint32_t)run;
    // which level do the codes in this leaf's records belong to?
    //   a candidate above (it was split, or this node would not exist): that candidate's level -> continue the chain;
    //   else this node is a predicted leaf, or an unsplit (first) candidate: its own level -> nothing to do;
    //   else (a non-candidate inner node of T'': the codes may belong to a candidate BELOW it): replay from the coordinates
    const bool cont = first_cand[k] != kNone;
    const bool replay = !cont && t.inner[i] && !t.candidate[i];
    if (cont) {
      r.cont_nodes.push_back(k);
      r.cont_from.push_back(first_cand[k]);
      r.cont_points += cnt[i];
    } else if (replay) {
      r.any_fix = true;
      r.fix_points += cnt[i];
      r.fix_nodes.push_back(k);
    } else if (t.inner[i]) {
      r.kept_points += cnt[i];
    }
    if (!t.inner[i]) {
      r.spec_map[t.leaf_rank[i]] = rank;  // the point's predicted leaf IS its leaf
      continue;
    }
    const uint32_t mapped = rank | (replay ? PCV_SPEC_MAP_REPLAY : 0u);
    size_t bp = 0;
    below[bp++] = i;
    while (bp) {
      const uint32_t j = below[--bp];
      if (!t.inner[j]) {
        r.spec_map[t.leaf_rank[j]] = mapped;
        if (replay) r.fix_level[t.leaf_rank[j]] = t.level[i];
      } else {
        for (unsigned c = 0; c < 8; ++c) below[bp++] = t.first_child[j] + c;
      }
    }
  }
  // inner nodes: [lo, hi) spans their leaves (children are contiguous and in digit order; bottom-up over the table)
  for (size_t k = nk; k-- > 0;) {
    if (!r.open[k]) continue;
    const uint32_t nchild = (uint32_t)__builtin_popcount(r.child_mask[k]);
    r.lo[k] = r.lo[r.first_child[k]];
    r.hi[k] = r.hi[r.first_child[k] + nchild - 1];
  }
  return PCV_SPEC_OK;
}

// ---- K6 work lists ---------------------------------------------------------------------------------------------------
uint32_t pcv_settle_items(const uint32_t* lo, const uint32_t* count, uint32_t num_leaves, PcvSettleItem* out) {
  uint32_t n = 0;
  for (uint32_t r = 0; r < num_leaves; ++r) {
    const uint64_t b0 = lo[r], e = b0 + count[r];
    for (uint64_t b = b0; b < e; b += kPcvSettleTile)
      out[n++] = PcvSettleItem{r, (uint32_t)b, (uint32_t)std::min<uint64_t>(b + kPcvSettleTile, e), 0u};
  }
  return n;
}

uint64_t pcv_climb_layout(const uint32_t* count, const uint8_t* climbs, uint32_t num_leaves, uint32_t* climb_base,
                          PcvSettleItem* out, uint32_t* num_items) {
  uint64_t total = 0;
  uint32_t n = 0;
  for (uint32_t r = 0; r < num_leaves; ++r) {
    climb_base[r] = (uint32_t)total;
    if (!climbs[r]) continue;
    const uint64_t k8 = ((uint64_t)count[r] + 7) / 8;
    for (uint64_t b = 0; b < k8; b += kPcvClimbTile)
      out[n++] = PcvSettleItem{r, (uint32_t)(total + b), (uint32_t)(total + std::min<uint64_t>(b + kPcvClimbTile, k8)),
                               (uint32_t)total /* == climb_base[r]: the leaf's first climber record */};
    total += k8;
  }
  *num_items = n;
  return total;
}

// CPU test hook (tests/test_spec_cpu.py): both lists for given leaf sizes; items come back as 4 x u32 each.
extern "C" int pcv_worklist_selftest(const uint32_t* lo, const uint32_t* count, const uint8_t* climbs, uint32_t num_leaves,
                                     uint32_t* settle_items, uint64_t* num_settle, uint32_t* climb_base, uint32_t* climb_items,
                                     uint64_t* num_climb, uint64_t* total_climbers) {
  static_assert(sizeof(PcvSettleItem) == 16, "work item");
  *num_settle = pcv_settle_items(lo, count, num_leaves, reinterpret_cast<PcvSettleItem*>(settle_items));
  uint32_t nc = 0;
  *total_climbers = pcv_climb_layout(count, climbs, num_leaves, climb_base, reinterpret_cast<PcvSettleItem*>(climb_items), &nc);
  *num_climb = nc;
  return 0;
}

// ---- CPU self-test hook (tests/test_spec_cpu.py) --------------------------------------------------------------------
// Runs the whole host logic on given full-depth path keys (from the oracle): strided sample -> sample tree (a plain
// CPU restatement of what the device node split produces) -> T'' -> walk every key down T'' -> exact counts -> true tree.
// Returns the status; the true tree comes back as (prefix, level, point count before promotion, open) per node.
extern "C" int pcv_spec_selftest(const uint64_t* keys, uint64_t n, uint32_t stride, uint32_t cap, double delta,
                                 double resolution, const double* edge, int nlevels, uint32_t force_mask,
                                 uint64_t node_capacity, uint64_t* out_prefix, uint8_t* out_level, uint64_t* out_count,
                                 uint8_t* out_open, uint64_t* out_num_nodes, uint64_t* out_stats /* [4] */) {
  PcvSpecParams p;
  p.cap = cap;
  p.resolution = resolution;
  p.edge = edge;
  p.nlevels = nlevels;
  p.force_mask = force_mask;
  p.scale = (double)stride;
  p.delta = delta;
  std::vector<uint64_t> sample;
  for (uint64_t i = 0; i < n; i += stride) sample.push_back(keys[i]);
  std::sort(sample.begin(), sample.end());
  const uint32_t thr = pcv_spec_sample_threshold(p);
  // sample tree exactly as pcv_launch_node_split lays it out
  std::vector<uint64_t> prefix{0};
  std::vector<uint32_t> lo{0}, hi{(uint32_t)sample.size()}, first{0};
  std::vector<uint8_t> level{0}, mask{0}, open{(uint8_t)(sample.empty() ? 0 : 1)};
  bool too_deep = false;
  for (size_t i = 0; i < prefix.size(); ++i) {
    if (!open[i]) continue;
    const int k = level[i] + 1;
    first[i] = (uint32_t)prefix.size();
    uint32_t b = lo[i];
    for (unsigned c = 0; c < 8; ++c) {
      const uint64_t next = c == 7 ? 0 : (prefix[i] | digit_bits(c + 1, k));
      const uint32_t e = c == 7 ? hi[i] : (uint32_t)(std::lower_bound(sample.begin() + b, sample.begin() + hi[i], next) - sample.begin());
      if (e > b) {
        mask[i] |= (uint8_t)(1u << c);
        bool op = (e - b) > thr && edge[k] > resolution;
        if (k == 1 && ((force_mask >> c) & 1u)) op = true;
        if (op && k >= nlevels) {
          too_deep = true;
          op = false;
        }
        prefix.push_back(prefix[i] | digit_bits(c, k));
        lo.push_back(b);
        hi.push_back(e);
        first.push_back(0);
        level.push_back((uint8_t)k);
        mask.push_back(0);
        open.push_back(op ? 1 : 0);
      }
      b = e;
    }
  }
  if (too_deep) return 100;
  PcvSampleTable st;
  st.num_nodes = (uint32_t)prefix.size();
  st.prefix = prefix.data();
  st.lo = lo.data();
  st.hi = hi.data();
  st.first_child = first.data();
  st.level = level.data();
  st.child_mask = mask.data();
  st.open = open.data();
  PcvSpecTree tree;
  pcv_spec_build_tree(p, st, &tree);
  std::vector<uint32_t> counts(tree.num_leaves, 0);
  std::vector<uint8_t> code_level(n);  // level of the codes the chain pass leaves in the point's record
  std::vector<uint32_t> pred_leaf(n);
  uint64_t kept = 0;
  for (uint64_t i = 0; i < n; ++i) {  // what the one chain pass does with the digits of the chain
    uint32_t rec = tree.walk[0];
    int l = 0, kl = 0;
    while (!(rec & PCV_SPEC_LEAF)) {
      if ((rec & PCV_SPEC_CANDIDATE) && kl == 0) kl = l;  // the first candidate on the path: its codes are what is kept
      ++l;
      rec = tree.walk[(rec & PCV_SPEC_INDEX_MASK) + (unsigned)((keys[i] >> (3 * (kKeyLevels - l))) & 7)];
    }
    kept += kl != 0;
    code_level[i] = (uint8_t)(kl ? kl : l);
    pred_leaf[i] = rec & PCV_SPEC_INDEX_MASK;
    ++counts[pred_leaf[i]];
  }
  PcvTrueTree tt;
  const PcvSpecStatus status = pcv_spec_resolve(p, tree, counts.data(), &tt);
  if (pcv_experiment("PCV_SPEC_TIME")) {  // host cost of the build's critical section (tools/spec_resolve_time.py)
    const auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < 200; ++rep) {
      PcvTrueTree again;
      pcv_spec_resolve(p, tree, counts.data(), &again);
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
    fprintf(stderr, "pcv_spec_resolve: %.1f us (T'' %zu nodes, %u leaves, true tree %zu nodes)\n", us, tree.prefix.size(),
            tree.num_leaves, tt.prefix.size());
  }
  if (out_stats) {
    out_stats[0] = tree.prefix.size();
    out_stats[1] = tree.num_leaves;
    out_stats[2] = kept;          // points that passed a candidate (their records carry that candidate's codes)
    out_stats[3] = tt.fix_points;  // points that replay the chain from their coordinates
  }
  if (status != PCV_SPEC_OK) return (int)status;
  {  // every point's record must carry codes the build knows how to turn into its leaf's codes
    std::vector<uint32_t> leaf_node;  // true-leaf rank -> node (depth-first order, as in pcv_spec_resolve)
    std::vector<uint32_t> st{0};
    while (!st.empty()) {
      const uint32_t k = st.back();
      st.pop_back();
      if (tt.open[k]) {
        const uint32_t nchild = (uint32_t)__builtin_popcount(tt.child_mask[k]);
        for (uint32_t c = nchild; c-- > 0;) st.push_back(tt.first_child[k] + c);
      } else {
        leaf_node.push_back(k);
      }
    }
    std::vector<int> from_level(tt.prefix.size(), -1);  // leaf node -> level the chain is continued from
    std::vector<uint8_t> replays(tt.prefix.size(), 0);
    for (size_t j = 0; j < tt.cont_nodes.size(); ++j) {
      const uint32_t leaf = tt.cont_nodes[j], from = tt.cont_from[j];
      if (!tt.open[from] || tt.level[from] >= tt.level[leaf]) return 103;
      // the candidate must be an ancestor of the leaf
      const int sh = 3 * (kKeyLevels - tt.level[from]);
      if (tt.level[from] && (tt.prefix[leaf] >> sh) != (tt.prefix[from] >> sh)) return 103;
      from_level[leaf] = tt.level[from];
    }
    for (uint32_t k : tt.fix_nodes) replays[k] = 1;
    uint64_t cont_seen = 0, fix_seen = 0;
    for (uint64_t i = 0; i < n; ++i) {
      const uint32_t m = tt.spec_map[pred_leaf[i]];
      const uint32_t leaf = leaf_node[m & PCV_SPEC_INDEX_MASK];
      // the point really lies in that leaf
      const int sh = 3 * (kKeyLevels - tt.level[leaf]);
      if (tt.level[leaf] && (keys[i] >> sh) != (tt.prefix[leaf] >> sh)) return 104;
      if (((m & PCV_SPEC_MAP_REPLAY) != 0) != (replays[leaf] != 0)) return 105;
      if (replays[leaf]) {
        ++fix_seen;  // codes ignored: replayed from the coordinates
      } else if (from_level[leaf] >= 0) {
        if ((int)code_level[i] != from_level[leaf]) return 106;  // continued from exactly the level the record holds
        ++cont_seen;
      } else if (code_level[i] != tt.level[leaf]) {
        return 107;  // nothing is done to this leaf: the record must already hold its level's codes
      }
    }
    if (cont_seen != tt.cont_points || fix_seen != tt.fix_points) return 108;
  }
  *out_num_nodes = tt.prefix.size();
  if (tt.prefix.size() > node_capacity) return 101;
  for (size_t k = 0; k < tt.prefix.size(); ++k) {
    out_prefix[k] = tt.prefix[k];
    out_level[k] = tt.level[k];
    out_count[k] = tt.hi[k] - tt.lo[k];
    out_open[k] = tt.open[k];
  }
  // the map must send every predicted leaf's points into the true leaf that spans them
  std::vector<uint64_t> per_leaf(tt.num_leaves, 0);
  for (uint32_t r = 0; r < tree.num_leaves; ++r) per_leaf[tt.spec_map[r] & PCV_SPEC_INDEX_MASK] += counts[r];
  uint32_t rank = 0;
  std::vector<uint32_t> stack{0};
  while (!stack.empty()) {
    const uint32_t k = stack.back();
    stack.pop_back();
    if (tt.open[k]) {
      const uint32_t nchild = (uint32_t)__builtin_popcount(tt.child_mask[k]);
      for (uint32_t c = nchild; c-- > 0;) stack.push_back(tt.first_child[k] + c);
    } else if (per_leaf[rank++] != (uint64_t)(tt.hi[k] - tt.lo[k])) {
      return 102;
    }
  }
  return 0;
}
