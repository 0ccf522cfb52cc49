// pcv_topology.hip — K4 node_split: the octree topology from the sorted path keys.
//
// Replaces the recursion of split()/should_split_node()/split_node() (reference
// src/octree/generation.rs:58-193): a child exists iff at least one point carries its path prefix, and a
// child is split again iff `count > MAX_POINTS_PER_NODE && child.edge > resolution` (generation.rs:128-150);
// the root is always split (generation.rs:312-323). With the keys sorted, a node is a contiguous range and its
// eight children are found with seven lower-bound searches.
//
// Per level two launches: (A) one wave64 per (open node, child digit) runs a 64-ary search — every lane probes
// one key, a ballot narrows the range 64x per step (5 dependent loads for 2^30 keys instead of 30);
// (B) one workgroup scans the child counts and appends the new level to the node table in (parent, digit)
// order, so the table is deterministic and BFS-ordered.
#include "pcv_internal.h"
#include "pcv_spec.h"

namespace {

enum { CNT_NODES = 0, CNT_ERROR = 1, CNT_LEVEL_START = 2, CNT_OPEN = 60 /* [60], [61]: entries of the two open-node lists */ };

// scratch layout of the two-levels-per-launch kernels (see split2_search_kernel below)
__device__ __forceinline__ uint32_t* pcv_split2_open_list(const PcvNodeTableDev& t, int which) {
  return t.bounds + (size_t)81 * t.max_open + (size_t)which * t.max_open;
}
__device__ __forceinline__ uint32_t* split2_mid_list(const PcvNodeTableDev& t) { return t.bounds + (size_t)83 * t.max_open; }

template <typename KeyT>
__device__ __forceinline__ uint32_t wave_lower_bound(const KeyT* __restrict__ keys, uint32_t lo, uint32_t hi,
                                                     KeyT target, int lane) {
  // invariant: every key before lo is < target, every key at or after hi is >= target
  while (hi - lo > 64) {
    const uint32_t span = hi - lo;
    const uint32_t step = (span + 63) / 64;
    uint64_t idx = (uint64_t)lo + (uint64_t)(lane + 1) * step - 1;
    const bool in = idx < hi;
    const KeyT v = keys[in ? idx : (uint64_t)hi - 1];
    const uint64_t m = __ballot(in && v < target);
    const uint32_t cnt = __popcll(m);  // probes are monotone: the mask is a prefix
    const uint64_t nlo = (uint64_t)lo + (uint64_t)cnt * step;
    const uint64_t nhi = (uint64_t)lo + (uint64_t)(cnt + 1) * step - 1;  // first probe that is >= target
    lo = (uint32_t)(nlo < hi ? nlo : hi);
    hi = (uint32_t)(nhi < hi ? nhi : hi);
  }
  const bool in = lo + (uint32_t)lane < hi;
  const KeyT v = in ? keys[lo + lane] : (KeyT)0;
  const uint64_t m = __ballot(in && v < target);
  return lo + (uint32_t)__popcll(m);
}

__global__ __launch_bounds__(256) void init_root_kernel(PcvNodeTableDev t, uint32_t n) {
  if (threadIdx.x == 0) {
    t.prefix[0] = 0;
    if (t.prefix_lo) t.prefix_lo[0] = 0;
    t.lo[0] = 0;
    t.hi[0] = n;
    t.parent[0] = 0xffffffffu;
    t.first_child[0] = 0;
    t.level[0] = 0;
    t.child_mask[0] = 0;
    t.open[0] = n > 0 ? 1 : 0;  // the root is always split (generation.rs:312-323)
    t.counters[CNT_NODES] = 1;
    t.counters[CNT_ERROR] = 0;
    t.counters[CNT_LEVEL_START + 0] = 0;
    t.counters[CNT_LEVEL_START + 1] = 1;
    // two levels per launch (split2_*): the open nodes of the level to expand, as a list
    t.counters[CNT_OPEN + 0] = n > 0 ? 1 : 0;
    t.counters[CNT_OPEN + 1] = 0;
    if (t.max_open) pcv_split2_open_list(t, 0)[0] = 0;
  }
}

// ---- two levels per launch -------------------------------------------------------------------------------------------------
// The sample tree of the single-chain build is ~12 levels of two launches each, every one of them 5-8 us of dispatch for
// microseconds of work. Two levels go through ONE search launch and ONE assign launch:
//   split2_search: one workgroup (8 waves) per OPEN node of level k - 1 (taken from a list, not found by scanning the
//     level): its seven child boundaries (seven 64-ary searches, as split_search), then — the counts are now known — the
//     seven boundaries of every child that could be split itself (count > capacity, or forced), 8 searches at a time;
//   split2_assign: one workgroup appends level k exactly as split_assign does, keeps the open ones among the new nodes as a
//     list in creation order, and appends level k + 1 from their boundaries; the open nodes of level k + 1 become the list
//     of the next launch pair.
// Same node table, same order (children in (parent, digit) order, levels breadth-first) as the one-level kernels.
// Scratch (t.bounds): [9 x max_open child bounds | 72 x max_open grandchild bounds | 2 x max_open list A | list B | 2 x
// max_open mid list], max_open = keys / capacity + 16 (an open node holds more than `capacity` keys; the root and the
// forced level-1 nodes are the + 16).

template <typename KeyT>
__global__ __launch_bounds__(512) void split2_search_kernel(PcvNodeTableDev t, const KeyT* __restrict__ keys, int k, int cur,
                                                             uint32_t max_points, uint32_t force_mask, int second_level) {
  __shared__ uint32_t sb[9];
  __shared__ uint32_t cand[8];
  __shared__ uint32_t ncand;
  const uint32_t count = t.counters[CNT_OPEN + cur];
  const uint32_t idx = blockIdx.x;
  if (idx >= count || idx >= t.max_open) return;  // workgroup-uniform
  const uint32_t node = pcv_split2_open_list(t, cur)[idx];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int shift = 3 * (PCV_MAX_KEY_LEVELS - k);
  const uint32_t lo = t.lo[node], hi = t.hi[node];
  const uint64_t pfx = t.prefix[node];
  if (wave == 0) {
    if (lane == 0) sb[0] = lo, sb[8] = hi, ncand = 0;
  } else {
    const uint32_t b = wave_lower_bound<KeyT>(keys, lo, hi, (KeyT)(pfx | ((uint64_t)wave << shift)), lane);
    if (lane == 0) sb[wave] = b;
  }
  __syncthreads();
  uint32_t* bd = t.bounds + (size_t)idx * 9u;
  if (threadIdx.x < 9) bd[threadIdx.x] = sb[threadIdx.x];
  if (!second_level) return;
  if (threadIdx.x == 0) {  // children that may be split themselves (a superset of what split2_assign will open)
    uint32_t m = 0;
    for (uint32_t c = 0; c < 8; ++c)
      if (sb[c + 1] - sb[c] > max_points || (k == 1 && ((force_mask >> c) & 1u) && sb[c + 1] > sb[c])) cand[m++] = c;
    ncand = m;
  }
  __syncthreads();
  const uint32_t tasks = ncand * 7u;
  uint32_t* bd2 = t.bounds + (size_t)9 * t.max_open + (size_t)idx * 72u;
  for (uint32_t task = wave; task < tasks; task += 8u) {  // wave-uniform
    const uint32_t c = cand[task / 7u], g = task % 7u + 1u;
    const uint32_t clo = sb[c], chi = sb[c + 1];
    const uint64_t target = pfx | ((uint64_t)c << shift) | ((uint64_t)g << (shift - 3));
    const uint32_t b = wave_lower_bound<KeyT>(keys, clo, chi, (KeyT)target, lane);
    if (lane == 0) {
      bd2[c * 9u + g] = b;
      if (g == 1) bd2[c * 9u] = clo, bd2[c * 9u + 8] = chi;
    }
  }
}

// one level appended by the whole workgroup: `src(i, b)` fills the 9 bounds of the i-th open parent and returns its node
// index; children are created in (parent, digit) order from index `running`; the open ones among them are listed (in
// creation order) in out_list with `out_src` = 8 x (parent's list position) + digit. Returns through shared memory.
struct Split2NoHook {
  __device__ __forceinline__ void operator()(uint32_t, uint32_t, uint32_t, int, bool, uint32_t) const {}
};
template <typename Src, typename Hook = Split2NoHook>
__device__ __forceinline__ void split2_append_level(const PcvNodeTableDev& t, const PcvLevels& lv, double resolution, uint32_t max_points,
                                                    int k, uint32_t force_mask, uint32_t parents, Src src, uint32_t* out_list,
                                                    uint32_t* out_src, uint32_t* wave_tot, uint32_t* wave_open, uint32_t* running,
                                                    uint32_t* running_open,
                                                    Hook hook = Hook() /* hook(child node, parent's list position, parent node, digit, open, its slot in out_list) */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int shift = 3 * (PCV_MAX_KEY_LEVELS - k);
  for (uint32_t chunk = 0; chunk < parents; chunk += 1024) {
    const uint32_t i = chunk + threadIdx.x;
    uint32_t b[9];
    uint32_t mask = 0, cnt = 0, omask = 0, ocnt = 0, node = 0;
    const bool active = i < parents;
    if (active) {
      node = src(i, b);
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (b[c + 1] > b[c]) {
          mask |= 1u << c;
          ++cnt;
          // should_split_node (generation.rs:128-150): count > MAX && child edge > resolution; multi-GPU build: a level-1
          // node that is split in the GLOBAL tree is split here too (PCV_BUILD_FORCE_SPLIT_L1)
          bool open = b[c + 1] - b[c] > max_points && lv.edge[k] > resolution;
          if (k == 1 && ((force_mask >> c) & 1u)) open = true;
          if (open && k >= lv.nlevels) {  // would need digits beyond the key width
            atomicOr(&t.counters[CNT_ERROR], 1u);
            open = false;
          }
          if (open) omask |= 1u << c, ++ocnt;
        }
    }
    uint32_t inc = cnt, oinc = ocnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(inc, o, 64), w = __shfl_up(oinc, o, 64);
      if (lane >= o) inc += v, oinc += w;
    }
    if (lane == 63) wave_tot[wave] = inc, wave_open[wave] = oinc;
    __syncthreads();
    uint32_t woff = 0, total = 0, ooff = 0, ototal = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const uint32_t v = wave_tot[w], u = wave_open[w];
      woff += (w < wave) ? v : 0u;
      total += v;
      ooff += (w < wave) ? u : 0u;
      ototal += u;
    }
    const uint32_t first = *running + woff + inc - cnt;
    uint32_t oslot = *running_open + ooff + oinc - ocnt;
    if (active) {
      t.first_child[node] = first;
      t.child_mask[node] = (uint8_t)mask;
      uint32_t j = first;
      const uint64_t pfx = t.prefix[node];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (!((mask >> c) & 1u)) continue;
        if (j < t.capacity) {
          const bool open = (omask >> c) & 1u;
          t.prefix[j] = pfx | ((uint64_t)c << shift);
          if (t.prefix_lo) t.prefix_lo[j] = 0;
          t.lo[j] = b[c];
          t.hi[j] = b[c + 1];
          t.parent[j] = node;
          t.first_child[j] = 0;
          t.level[j] = (uint8_t)k;
          t.child_mask[j] = 0;
          t.open[j] = open ? 1 : 0;
          hook(j, i, node, c, open, oslot);
          if (open) {
            if (oslot < t.max_open) {
              out_list[oslot] = j;
              if (out_src) out_src[oslot] = i * 8u + (uint32_t)c;
            } else {
              atomicOr(&t.counters[CNT_ERROR], 2u);  // (cannot happen: an open node holds more than `capacity` keys)
            }
            ++oslot;
          }
        } else {
          atomicOr(&t.counters[CNT_ERROR], 2u);  // node table capacity exceeded
          if ((omask >> c) & 1u) ++oslot;
        }
        ++j;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) *running += total, *running_open += ototal;
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void split2_assign_kernel(PcvNodeTableDev t, PcvLevels lv, double resolution, uint32_t max_points,
                                                              int k, int cur, uint32_t force_mask, int second_level) {
  __shared__ uint32_t wave_tot[16], wave_open[16];
  __shared__ uint32_t running, running_open;
  const uint32_t parents0 = t.counters[CNT_OPEN + cur] < t.max_open ? t.counters[CNT_OPEN + cur] : t.max_open;
  const uint32_t* cur_list = pcv_split2_open_list(t, cur);
  uint32_t* next_list = pcv_split2_open_list(t, cur ^ 1);
  uint32_t* mid = split2_mid_list(t);  // [max_open] nodes, [max_open] sources
  if (threadIdx.x == 0) running = t.counters[CNT_NODES], running_open = 0;
  __syncthreads();
  // level k from the child bounds of the open nodes of level k - 1
  split2_append_level(
      t, lv, resolution, max_points, k, force_mask, parents0,
      [&](uint32_t i, uint32_t (&b)[9]) {
        const uint32_t* bd = t.bounds + (size_t)i * 9u;
#pragma unroll
        for (int c = 0; c < 9; ++c) b[c] = bd[c];
        return cur_list[i];
      },
      second_level ? mid : next_list, second_level ? mid + t.max_open : nullptr, wave_tot, wave_open, &running, &running_open);
  __syncthreads();
  const uint32_t level_k_end = running < t.capacity ? running : t.capacity;
  const uint32_t open_k = running_open < t.max_open ? running_open : t.max_open;
  __syncthreads();
  if (!second_level) {
    if (threadIdx.x == 0) {
      t.counters[CNT_NODES] = level_k_end;
      t.counters[CNT_LEVEL_START + k + 1] = level_k_end;
      t.counters[CNT_OPEN + (cur ^ 1)] = open_k;
    }
    return;
  }
  __threadfence_block();  // the mid list was written by this workgroup
  if (threadIdx.x == 0) running_open = 0;
  __syncthreads();
  // level k + 1 from the grandchild bounds of the open nodes of level k
  split2_append_level(
      t, lv, resolution, max_points, k + 1, 0u, open_k,
      [&](uint32_t i, uint32_t (&b)[9]) {
        const uint32_t s = mid[t.max_open + i];  // 8 x (position of the parent's parent in the list) + digit
        const uint32_t* bd = t.bounds + (size_t)9 * t.max_open + (size_t)(s >> 3) * 72u + (size_t)(s & 7u) * 9u;
#pragma unroll
        for (int c = 0; c < 9; ++c) b[c] = bd[c];
        return mid[i];
      },
      next_list, nullptr, wave_tot, wave_open, &running, &running_open);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t total_nodes = running < t.capacity ? running : t.capacity;
    t.counters[CNT_NODES] = total_nodes;
    t.counters[CNT_LEVEL_START + k + 1] = level_k_end;
    t.counters[CNT_LEVEL_START + k + 2] = total_nodes;
    t.counters[CNT_OPEN + (cur ^ 1)] = running_open < t.max_open ? running_open : t.max_open;
  }
}

#ifdef PCV_EXPERIMENTS  // measured slower than sort + split (see the note at the call site, pcv_build.hip): experiment library only
// ---- the sample tree by COUNTING (round 5) -----------------------------------------------------------------------------------
// MEASURED AND DROPPED (2-3 x slower than what it replaces: global atomics). The idea:
// The sample tree of the single-chain build comes out of a five-pass key sort (15 launches) and a node split by binary search
// in the sorted keys (12 launches): 0.28 ms of dependent 5-17 us launches for a table of a few thousand nodes. All the split
// needs from the keys is HOW MANY of them carry each prefix the tree opens, so the keys are counted instead, three levels per
// round (`group` g = levels 3 g + 1 .. 3 g + 3):
//   sample_count_kernel   every key finds the open level-3g node above it — `slot` s — by g look-ups through the slot maps of
//                         the groups above (map[g'][s' x 512 + the key's nine bits of group g'] = slot of the open node, or
//                         NONE: nothing below a closed node is counted) and bumps its three counters of the slot: 8 + 64 + 512
//                         per slot, global atomics (group 0: one slot, counted in LDS first). The same launch clears the
//                         counters and the map of the NEXT group (they are only as big as the open nodes a level can have);
//   sample_tree_kernel    ONE workgroup appends the group's three levels with split2_append_level — the code that lays out
//                         the table from sorted keys: same (parent, digit) order, same open / forced / too-deep rules — taking
//                         every child's count from the counters, and numbers the open nodes of level 3 g + 3 as the next
//                         group's slots.
// 2 x ceil(levels / 3) launches instead of 27, no sort. The node table is the one pcv_launch_node_split builds, except that
// lo / hi hold running counts instead of positions in a sorted array (the predicted tree only takes hi - lo).
constexpr uint32_t kSampleNone = 0xffffffffu;
constexpr uint32_t kSlotCounters = 8 + 64 + 512;  // per slot: counts of the children, grandchildren, great-grandchildren
struct SampleCountTables {
  uint32_t* counts;  // [groups][max_open][kSlotCounters]
  uint32_t* maps;    // [groups][max_open][512]
  uint32_t* nslot;   // per node: the slot of the open group node above it (or its own, at a group boundary) ...
  uint32_t* nq;      // ... and its digits below that node (0 .. 63)
  uint32_t max_open;
};
__device__ __forceinline__ uint32_t* sample_counts(const SampleCountTables& c, int g) { return c.counts + (size_t)g * c.max_open * kSlotCounters; }
__device__ __forceinline__ uint32_t* sample_map(const SampleCountTables& c, int g) { return c.maps + (size_t)g * c.max_open * 512u; }

// one counter bump per group of lanes that hold the same counter (the sample keys of a clustered cloud crowd into a few
// cells: left to themselves, 1.5 M keys put 10^5 atomics on ONE address of group 1, and same-address atomics retire one
// after the other — 0.54 ms for that launch). The leader also skips the bump once the counter has passed `sat`: the
// split only asks "more than the threshold?" and "inside the band?", a count beyond both may stop growing.
__device__ __forceinline__ void sample_bump_grouped(uint32_t* __restrict__ cnt, uint32_t idx, bool valid, uint32_t sat, int lane) {
  // every lane looks at its own counter first (one load latency for the wave, a stale value only costs a bump too many);
  // the grouping loop below is register work only
  const bool want = valid && cnt[idx] <= sat;
  uint64_t rem = __ballot(want);
  while (rem) {  // wave-uniform
    const int leader = (int)__builtin_ctzll(rem);
    const uint32_t i0 = (uint32_t)__builtin_amdgcn_readlane((int)idx, leader);
    const uint64_t same = __ballot(want && idx == i0);
    if (lane == leader) atomicAdd(&cnt[i0], (uint32_t)__popcll(same));
    rem &= ~same;
  }
}

__global__ __launch_bounds__(256) void sample_count_kernel(const uint64_t* __restrict__ keys, uint32_t n, int g, int groups, uint32_t sat,
                                                            SampleCountTables c) {
  __shared__ uint32_t hist[kSlotCounters];
  // the next group's tables (group 0's are cleared by sample_tree_init_kernel)
  if (g + 1 < groups) {
    uint32_t* nc = sample_counts(c, g + 1);
    uint32_t* nm = sample_map(c, g + 1);
    const size_t total_c = (size_t)c.max_open * kSlotCounters, total_m = (size_t)c.max_open * 512u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total_c; i += (size_t)gridDim.x * 256) nc[i] = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total_m; i += (size_t)gridDim.x * 256) nm[i] = kSampleNone;
  }
  uint32_t* cnt = sample_counts(c, g);
  const int lane = threadIdx.x & 63;
  if (g == 0) {
    for (uint32_t i = threadIdx.x; i < kSlotCounters; i += 256) hist[i] = 0;
    __syncthreads();
  }
  const int sh = 3 * (PCV_MAX_KEY_LEVELS - 3 * (g + 1));  // >= 0: at most 7 groups of three levels in a key word
  const uint32_t per = (n + gridDim.x * 256u - 1u) / (gridDim.x * 256u);  // the same trip count for every lane (the ballots)
  for (uint32_t it = 0; it < per; ++it) {
    const uint32_t i = (it * gridDim.x + blockIdx.x) * 256u + threadIdx.x;
    const bool in = i < n;
    const uint64_t key = in ? keys[i] : 0ull;
    uint32_t s = in ? 0u : kSampleNone;
    for (int q = 0; q < g; ++q) {
      const uint32_t bits = (uint32_t)(key >> (3 * (PCV_MAX_KEY_LEVELS - 3 * (q + 1)))) & 511u;
      if (s != kSampleNone) s = sample_map(c, q)[(size_t)s * 512u + bits];
    }
    const bool valid = s != kSampleNone;
    const uint32_t bits = (uint32_t)(key >> sh) & 511u;
    if (g == 0) {
      if (valid) {
        atomicAdd(&hist[bits >> 6], 1u);
        atomicAdd(&hist[8 + (bits >> 3)], 1u);
        atomicAdd(&hist[72 + bits], 1u);
      }
    } else {
      const uint32_t row = valid ? s * kSlotCounters : 0u;
      sample_bump_grouped(cnt, row + (bits >> 6), valid, sat, lane);
      sample_bump_grouped(cnt, row + 8u + (bits >> 3), valid, sat, lane);
      if (valid && cnt[row + 72u + bits] <= sat) atomicAdd(&cnt[row + 72u + bits], 1u);  // 512 per slot: mostly one lane each
    }
  }
  if (g == 0) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kSlotCounters; i += 256)
      if (hist[i]) atomicAdd(&cnt[i], hist[i]);
  }
}

// the root (init_root_kernel) + group 0's counters and map
__global__ __launch_bounds__(256) void sample_tree_init_kernel(PcvNodeTableDev t, uint32_t n, SampleCountTables c) {
  for (uint32_t i = threadIdx.x; i < kSlotCounters; i += 256) sample_counts(c, 0)[i] = 0;
  for (uint32_t i = threadIdx.x; i < 512; i += 256) sample_map(c, 0)[i] = kSampleNone;
  if (threadIdx.x == 0) {
    t.prefix[0] = 0;
    if (t.prefix_lo) t.prefix_lo[0] = 0;
    t.lo[0] = 0;
    t.hi[0] = n;
    t.parent[0] = 0xffffffffu;
    t.first_child[0] = 0;
    t.level[0] = 0;
    t.child_mask[0] = 0;
    t.open[0] = n > 0 ? 1 : 0;
    t.counters[CNT_NODES] = 1;
    t.counters[CNT_ERROR] = 0;
    t.counters[CNT_LEVEL_START + 0] = 0;
    t.counters[CNT_LEVEL_START + 1] = 1;
    t.counters[CNT_OPEN + 0] = n > 0 ? 1 : 0;
    t.counters[CNT_OPEN + 1] = 0;
    pcv_split2_open_list(t, 0)[0] = 0;
    c.nslot[0] = 0;
    c.nq[0] = 0;
  }
}

__global__ __launch_bounds__(1024) void sample_tree_kernel(PcvNodeTableDev t, PcvLevels lv, double resolution, uint32_t max_points, int g,
                                                            int cur /* which open list holds the open nodes of level 3 g */,
                                                            uint32_t force_mask, SampleCountTables c) {
  __shared__ uint32_t wave_tot[16], wave_open[16];
  __shared__ uint32_t running, running_open;
  const uint32_t* cnt = sample_counts(c, g);
  uint32_t* map = sample_map(c, g);
  if (threadIdx.x == 0) running = t.counters[CNT_NODES];
  __syncthreads();
  for (int r = 0; r < 3; ++r) {
    const int k = 3 * g + r + 1;
    if (k > lv.nlevels) break;  // (workgroup-uniform)
    const uint32_t parents = t.counters[CNT_OPEN + cur] < t.max_open ? t.counters[CNT_OPEN + cur] : t.max_open;
    const uint32_t* cur_list = pcv_split2_open_list(t, cur);
    uint32_t* next_list = pcv_split2_open_list(t, cur ^ 1);
    const uint32_t off = r == 0 ? 0u : r == 1 ? 8u : 72u;
    if (threadIdx.x == 0) running_open = 0;
    __syncthreads();
    split2_append_level(
        t, lv, resolution, max_points, k, force_mask, parents,
        [&](uint32_t i, uint32_t (&b)[9]) {
          const uint32_t node = cur_list[i];
          const uint32_t* row = cnt + (size_t)c.nslot[node] * kSlotCounters + off + c.nq[node] * 8u;
          b[0] = 0;
#pragma unroll
          for (int ch = 0; ch < 8; ++ch) b[ch + 1] = b[ch] + row[ch];
          if (r == 2) {  // below closed or missing children nothing is counted in the next group
            uint32_t* m = map + (size_t)c.nslot[node] * 512u + c.nq[node] * 8u;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) m[ch] = kSampleNone;
          }
          return node;
        },
        next_list, nullptr, wave_tot, wave_open, &running, &running_open,
        [&](uint32_t j, uint32_t, uint32_t node, int ch, bool open, uint32_t oslot) {
          const uint32_t s = c.nslot[node], q = c.nq[node] * 8u + (uint32_t)ch;
          if (r < 2) {
            c.nslot[j] = s;
            c.nq[j] = q;
          } else {  // a group boundary: the open node is a slot of the next group (its position among the level's open nodes)
            c.nslot[j] = open && oslot < t.max_open ? oslot : 0u;
            c.nq[j] = 0;
            if (open && oslot < t.max_open) map[(size_t)s * 512u + q] = oslot;
          }
        });
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t end = running < t.capacity ? running : t.capacity;
      t.counters[CNT_NODES] = end;
      t.counters[CNT_LEVEL_START + k + 1] = end;
      t.counters[CNT_OPEN + (cur ^ 1)] = running_open < t.max_open ? running_open : t.max_open;
    }
    __threadfence_block();
    __syncthreads();  // the next level reads the list and the counters this one wrote (same workgroup)
    cur ^= 1;
  }
}

#endif  // PCV_EXPERIMENTS

// (A) child boundaries of every open node of level k-1.
template <typename KeyT>
__global__ __launch_bounds__(256) void split_search_kernel(PcvNodeTableDev t, const KeyT* __restrict__ keys, int k,
                                                            const uint64_t* __restrict__ keys_lo) {
  const uint32_t begin = t.counters[CNT_LEVEL_START + k - 1];
  const uint32_t end = t.counters[CNT_LEVEL_START + k];
  const int lane = threadIdx.x & 63;
  const uint32_t wave_global = (blockIdx.x * 256u + threadIdx.x) >> 6;
  const uint32_t nwaves = gridDim.x * 4u;
  const int shift = 3 * (PCV_MAX_KEY_LEVELS - k);
  const uint32_t items = (end - begin) * 7u;
  for (uint32_t it = wave_global; it < items; it += nwaves) {
    const uint32_t node = begin + it / 7u;
    const uint32_t c = it % 7u + 1u;
    if (!t.open[node]) continue;  // wave-uniform
    const uint32_t lo = t.lo[node], hi = t.hi[node];
    uint32_t b;
    if (k > PCV_MAX_KEY_LEVELS) {
      // deep tree: every key of this node has the same first word; the children are told apart by the second
      const uint64_t target = t.prefix_lo[node] | ((uint64_t)c << (3 * (2 * PCV_MAX_KEY_LEVELS - k)));
      b = wave_lower_bound<uint64_t>(keys_lo, lo, hi, target, lane);
    } else {
      const uint64_t target64 = t.prefix[node] | ((uint64_t)c << shift);
      const KeyT target = sizeof(KeyT) == 8 ? (KeyT)target64 : (KeyT)(target64 >> 33);  // u32 keys hold key >> 33
      b = wave_lower_bound<KeyT>(keys, lo, hi, target, lane);
    }
    if (lane == 0) {
      uint32_t* bd = t.bounds + (uint64_t)(node - begin) * 9u;
      bd[c] = b;
      if (c == 1) {
        bd[0] = lo;
        bd[8] = hi;
      }
    }
  }
}

// (B) append level k. One workgroup; nodes of level k-1 are visited in order, children get consecutive
// indices in (parent, digit) order.
__global__ __launch_bounds__(1024) void split_assign_kernel(PcvNodeTableDev t, PcvLevels lv, double resolution,
                                                             uint32_t max_points, int k, uint32_t force_mask) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t running;
  const uint32_t begin = t.counters[CNT_LEVEL_START + k - 1];
  const uint32_t end = t.counters[CNT_LEVEL_START + k];
  const uint32_t base0 = t.counters[CNT_NODES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int shift = 3 * (PCV_MAX_KEY_LEVELS - k);
  if (threadIdx.x == 0) running = base0;
  __syncthreads();
  for (uint32_t chunk = begin; chunk < end; chunk += 1024) {
    const uint32_t node = chunk + threadIdx.x;
    uint32_t b[9];
    uint32_t mask = 0, cnt = 0;
    const bool active = node < end && t.open[node];
    if (active) {
      const uint32_t* bd = t.bounds + (uint64_t)(node - begin) * 9u;
#pragma unroll
      for (int c = 0; c < 9; ++c) b[c] = bd[c];
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (b[c + 1] > b[c]) {
          mask |= 1u << c;
          ++cnt;
        }
    }
    uint32_t inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      uint32_t v = wave_tot[w];
      woff += (w < wave) ? v : 0u;
      total += v;
    }
    const uint32_t first = running + woff + inc - cnt;
    if (active) {
      t.first_child[node] = first;
      t.child_mask[node] = (uint8_t)mask;
      uint32_t j = first;
      const uint64_t pfx = t.prefix[node];
      const uint64_t pfx_lo = t.prefix_lo ? t.prefix_lo[node] : 0ull;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (!((mask >> c) & 1u)) continue;
        if (j < t.capacity) {
          const uint32_t count = b[c + 1] - b[c];
          // should_split_node (generation.rs:128-150): count > MAX && child edge > resolution
          bool open = count > max_points && lv.edge[k] > resolution;
          // multi-GPU build: a level-1 node that is split in the GLOBAL tree is split here too, whatever share of
          // its points this rank holds (pcv_build_params.flags, PCV_BUILD_FORCE_SPLIT_L1)
          if (k == 1 && ((force_mask >> c) & 1u)) open = true;
          if (open && k >= lv.nlevels) {
            // would need digits beyond the key width
            atomicOr(&t.counters[CNT_ERROR], 1u);
            open = false;
          }
          if (k > PCV_MAX_KEY_LEVELS) {
            t.prefix[j] = pfx;
            t.prefix_lo[j] = pfx_lo | ((uint64_t)c << (3 * (2 * PCV_MAX_KEY_LEVELS - k)));
          } else {
            t.prefix[j] = pfx | ((uint64_t)c << shift);
            if (t.prefix_lo) t.prefix_lo[j] = 0;
          }
          t.lo[j] = b[c];
          t.hi[j] = b[c + 1];
          t.parent[j] = node;
          t.first_child[j] = 0;
          t.level[j] = (uint8_t)k;
          t.child_mask[j] = 0;
          t.open[j] = open ? 1 : 0;
        } else {
          atomicOr(&t.counters[CNT_ERROR], 2u);  // node table capacity exceeded
        }
        ++j;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) running += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    uint32_t total_nodes = running < t.capacity ? running : t.capacity;
    t.counters[CNT_NODES] = total_nodes;
    t.counters[CNT_LEVEL_START + k + 1] = total_nodes;
  }
}

// counters + nodes into one contiguous block (one D2H copy instead of eight)
__global__ __launch_bounds__(256) void pack_node_table_kernel(PcvNodeTableDev t, uint8_t* __restrict__ packed) {
  const uint32_t count = t.counters[CNT_NODES] < t.capacity ? t.counters[CNT_NODES] : t.capacity;
  if (blockIdx.x == 0 && threadIdx.x < 64) reinterpret_cast<uint32_t*>(packed)[threadIdx.x] = t.counters[threadIdx.x];
  PcvPackedNode* out = reinterpret_cast<PcvPackedNode*>(packed + kPcvPackHeader);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
    PcvPackedNode n;
    n.prefix = t.prefix[i];
    n.lo = t.lo[i];
    n.hi = t.hi[i];
    n.first_child = t.first_child[i];
    n.level = t.level[i];
    n.child_mask = t.child_mask[i];
    n.open = t.open[i];
    n.pad = 0;
    out[i] = n;
  }
}

// ---- single-chain build: the predicted tree T'' built ON THE DEVICE from the sample's node table ------------------------
// (what pcv_spec_build_tree does on the host, same numbering: the k-th open sample node — table order — owns the T''
// nodes 1 + 8 k .. 1 + 8 k + 7, a leaf is named by its own index). With the walk table on the device the one chain
// pass starts without a host round trip; the host mirrors the tree (pcv_spec_tree_from_walk) while that pass runs.
// info: [0] number of T'' nodes, [1] error flags of the sample split, [2] sample nodes, [3] any candidate
__global__ __launch_bounds__(1024) void spec_tree_scan_kernel(PcvNodeTableDev t, uint32_t* __restrict__ ord,
                                                               uint32_t* __restrict__ info, uint32_t* __restrict__ pool_ctr) {
  static_assert(kPcvPoolRegions <= 1024, "one counter per lane");
  // the chain pass's pool counters (entries of `wide` handed out per region, pcv_encode.hip chain_pass_kernel): zero before every pass
  if (pool_ctr && threadIdx.x < kPcvPoolRegions) pool_ctr[threadIdx.x] = 0;
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t running;
  const uint32_t count = t.counters[CNT_NODES] < t.capacity ? t.counters[CNT_NODES] : t.capacity;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  for (uint32_t base = 0; base < count; base += 1024) {
    const uint32_t idx = base + threadIdx.x;
    const uint32_t v = idx < count ? (uint32_t)t.open[idx] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      woff += (w < wave) ? wave_tot[w] : 0u;
      tot += wave_tot[w];
    }
    if (idx < count) ord[idx] = running + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 0) running += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    info[0] = 1u + 8u * running;
    info[1] = t.counters[CNT_ERROR];
    info[2] = count;
    info[3] = 0;
  }
}

__global__ __launch_bounds__(256) void spec_tree_emit_kernel(PcvNodeTableDev t, const uint32_t* __restrict__ ord, double upper,
                                                              uint32_t force_mask, uint32_t* __restrict__ walk,
                                                              uint32_t* __restrict__ sparent, uint8_t* __restrict__ slevel,
                                                              uint32_t* __restrict__ info) {
  const uint32_t count = t.counters[CNT_NODES] < t.capacity ? t.counters[CNT_NODES] : t.capacity;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
    if (!t.open[i]) continue;
    const uint32_t level = t.level[i];
    const uint32_t digit = level ? (uint32_t)(t.prefix[i] >> (3 * (PCV_MAX_KEY_LEVELS - level))) & 7u : 0u;
    const uint32_t id = i == 0 ? 0u : 1u + 8u * ord[t.parent[i]] + digit;
    const uint32_t base = 1u + 8u * ord[i];
    if (i == 0) {  // the root is always split and never a candidate
      walk[0] = base;
      sparent[0] = 0xffffffffu;
      slevel[0] = 0;
    }
    const uint32_t mask = t.child_mask[i];
    uint32_t next = t.first_child[i];
    bool cand_seen = false;
#pragma unroll
    for (uint32_t c = 0; c < 8; ++c) {
      const uint32_t child = base + c;
      uint32_t rec = child | PCV_SPEC_LEAF;
      if ((mask >> c) & 1u) {
        const uint32_t j = next++;
        if (t.open[j]) {
          const bool forced = level == 0 && ((force_mask >> c) & 1u);  // level-1 nodes the global tree splits anyway
          const bool cand = !forced && (double)(t.hi[j] - t.lo[j]) <= upper;
          rec = (1u + 8u * ord[j]) | (cand ? PCV_SPEC_CANDIDATE : 0u);
          cand_seen = cand_seen || cand;
        }
      }
      walk[child] = rec;
      sparent[child] = id;
      slevel[child] = (uint8_t)(level + 1);
    }
    if (cand_seen) atomicOr(&info[3], 1u);
  }
}


// ---- single-chain build: the rank map ON THE DEVICE (what pcv_spec_resolve computes on the host, pcv_spec.cpp) ----------
// Exact counts per predicted leaf in, predicted-leaf -> true-leaf map out, so that the record sort can start without the
// counts travelling to the host and the map travelling back (the host still derives the true tree from the same counts
// — it needs the node tables — but beside the sort, not in front of it). One workgroup; T'' is level-major (the children
// of the k-th open sample node, table order, sit at 1 + 8 k ..), so every level is a contiguous index range:
//   bottom-up: exact count of every node, "would split" (should_split_node, generation.rs:128-150, with the exact count),
//              true leaves in the subtree;
//   top-down : depth-first rank of every true leaf = rank base of its parent + leaves of its earlier siblings (children in
//              digit order: the order pcv_spec_resolve and pcv_build_finish rank them in); everything below a true leaf
//              inherits its rank; PCV_SPEC_MAP_REPLAY where the leaf is a non-candidate inner node of T'' with no
//              candidate above it.
// Both sides apply the same integer rules to the same counts; a disagreement would surface as a parity failure (the
// device map drives the sort, the host's tree the tables).
// kLds: the four per-node arrays (walk record, count, leaves + state, rank base) live in LDS — a predicted tree of up to
// ~10 000 nodes (100-150 M points at the default capacity) fits the 160 KB, and each of the ~26 level steps then costs
// LDS latency instead of two or three dependent trips to L2 (88 us -> ~20 us for the 7 489 nodes of the bench tree).
template <bool kLds>
__global__ __launch_bounds__(1024) void spec_resolve_kernel(PcvLevels lv, double resolution, uint32_t cap, uint32_t force_mask,
                                                             const uint32_t* __restrict__ g_walk, const uint8_t* __restrict__ slevel,
                                                             uint32_t tn, uint32_t* __restrict__ g_cnt, uint32_t* __restrict__ g_nst,
                                                             uint32_t* __restrict__ g_base, uint32_t* __restrict__ map,
                                                             uint32_t* __restrict__ out /* [0] true leaves, [1] 1 = too shallow */) {
  extern __shared__ uint32_t dyn[];
  __shared__ uint32_t lstart[PCV_MAX_KEY_LEVELS + 3];
  __shared__ uint32_t too_shallow;
  // generic pointers: LDS (flat access) or global memory
  const uint32_t* walk = kLds ? dyn : g_walk;
  uint32_t* cnt = kLds ? dyn + tn : g_cnt;
  uint32_t* nst = kLds ? dyn + 2 * (size_t)tn : g_nst;   // leaves in the subtree (bits 0..27) | state (bits 28..30, see below)
  uint32_t* base = kLds ? dyn + 3 * (size_t)tn : g_base;
  const uint32_t t = threadIdx.x;
  if (t < PCV_MAX_KEY_LEVELS + 3) lstart[t] = tn;
  if (t == 0) too_shallow = 0;
  if (kLds)
    for (uint32_t i = t; i < tn; i += 1024) {
      dyn[i] = g_walk[i];
      dyn[tn + i] = g_cnt[i];
    }
  __syncthreads();
  for (uint32_t i = t; i < tn; i += 1024)
    if (i == 0 || slevel[i] != slevel[i - 1]) lstart[slevel[i]] = i;
  __syncthreads();
  int maxl = 0;
  for (int l = 0; l <= PCV_MAX_KEY_LEVELS; ++l)
    if (lstart[l] < tn) maxl = l;
  const uint32_t l1_first = walk[0] & PCV_SPEC_INDEX_MASK;  // the root's children (level 1) start here
  auto would_split = [&](uint32_t i, int level, uint32_t c) -> bool {
    if (level == 0) return true;  // the root is always split (generation.rs:312-323)
    if (level == 1 && ((force_mask >> ((i - l1_first) & 7u)) & 1u)) return true;  // multi-GPU build: the global tree splits it
    return c > cap && lv.edge[level] > resolution;
  };
  constexpr uint32_t kLeavesMask = 0x0fffffffu;
  for (int L = maxl; L >= 0; --L) {  // bottom-up
    const uint32_t b = lstart[L], e = lstart[L + 1];
    for (uint32_t i = b + t; i < e; i += 1024) {
      const uint32_t rec = walk[i];
      uint32_t c, nl;
      if (rec & PCV_SPEC_LEAF) {
        c = cnt[i];
        // the prediction stops above where the tree goes on (an empty octant is no node at all, forced or not: the host's
        // resolve never visits it, pcv_spec.cpp)
        if (c > 0 && would_split(i, L, c)) atomicOr(&too_shallow, 1u);
        nl = c > 0 ? 1u : 0u;
      } else {
        const uint32_t first = rec & PCV_SPEC_INDEX_MASK;
        uint32_t cs[8], ns[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          cs[k] = cnt[first + k];
          ns[k] = nst[first + k] & kLeavesMask;
        }
        c = ((cs[0] + cs[1]) + (cs[2] + cs[3])) + ((cs[4] + cs[5]) + (cs[6] + cs[7]));
        cnt[i] = c;
        nl = would_split(i, L, c) ? ((ns[0] + ns[1]) + (ns[2] + ns[3])) + ((ns[4] + ns[5]) + (ns[6] + ns[7])) : (c > 0 ? 1u : 0u);
      }
      nst[i] = nl;
    }
    __syncthreads();
  }
  // top-down. state (bits 28..29 of nst): 0 = no point below (or under such a node), 1 = inner node of the true tree,
  // 2 = a true leaf or below one; bit 30: a candidate node lies above on the path
  if (t == 0) {
    nst[0] = (nst[0] & kLeavesMask) | (1u << 28);
    base[0] = 0;
  }
  __syncthreads();
  for (int L = 0; L <= maxl; ++L) {
    const uint32_t b = lstart[L], e = lstart[L + 1];
    for (uint32_t i = b + t; i < e; i += 1024) {
      const uint32_t rec = walk[i];
      const uint32_t s = nst[i] >> 28;
      if (rec & PCV_SPEC_LEAF) {
        map[i] = (s & 3u) == 2u ? base[i] : 0u;
        continue;
      }
      const uint32_t first = rec & PCV_SPEC_INDEX_MASK;
      if ((s & 3u) == 1u) {
        uint32_t run = base[i];
        const uint32_t ca = ((s >> 2) & 1u) | ((rec & PCV_SPEC_CANDIDATE) ? 1u : 0u);
        uint32_t cs[8], ns[8], ws[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // every load of the eight children before anything depends on one
          cs[k] = cnt[first + k];
          ns[k] = nst[first + k] & kLeavesMask;
          ws[k] = walk[first + k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t ch = first + k;
          if (cs[k] == 0) {
            nst[ch] = ns[k];  // state 0
          } else if (would_split(ch, L + 1, cs[k])) {
            nst[ch] = ns[k] | ((1u | (ca << 2)) << 28);
            base[ch] = run;
            run += ns[k];
          } else {
            const bool replay = !ca && !(ws[k] & PCV_SPEC_LEAF) && !(ws[k] & PCV_SPEC_CANDIDATE);
            nst[ch] = ns[k] | (2u << 28);
            base[ch] = run | (replay ? PCV_SPEC_MAP_REPLAY : 0u);
            run += 1;
          }
        }
      } else {  // nothing to decide below a true leaf (or below an empty node): the children inherit
        const uint32_t v = base[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          nst[first + k] = (nst[first + k] & kLeavesMask) | (s << 28);
          base[first + k] = v;
        }
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    out[0] = nst[0] & kLeavesMask;
    out[1] = too_shallow;
  }
}

}  // namespace

void pcv_launch_spec_resolve(pcv_ctx* ctx, const PcvLevels& lv, double resolution, uint32_t cap, uint32_t force_mask, const uint32_t* walk,
                             const uint8_t* slevel, uint32_t tn, uint32_t* counts, uint32_t* nst, uint32_t* base, uint32_t* map,
                             uint32_t* out) {
  // small trees: the per-node arrays in LDS (16 bytes per node; the opt-in for more than 64 KB of dynamic LDS is made once)
  constexpr size_t kLdsBudget = 156 * 1024;
  static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&spec_resolve_kernel<true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget) == hipSuccess;
  const size_t need = (size_t)tn * 16;
  if (lds_ok && need <= kLdsBudget)
    hipLaunchKernelGGL((spec_resolve_kernel<true>), dim3(1), dim3(1024), need, ctx->stream, lv, resolution, cap, force_mask, walk, slevel, tn,
                       counts, nst, base, map, out);
  else
    hipLaunchKernelGGL((spec_resolve_kernel<false>), dim3(1), dim3(1024), 0, ctx->stream, lv, resolution, cap, force_mask, walk, slevel, tn,
                       counts, nst, base, map, out);
}

void pcv_launch_spec_tree(pcv_ctx* ctx, const PcvNodeTableDev& t, double upper, uint32_t force_mask, uint32_t* ord, uint32_t* walk,
                          uint32_t* sparent, uint8_t* slevel, uint32_t* info, uint32_t* pool_ctr) {
  hipLaunchKernelGGL(spec_tree_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, t, ord, info, pool_ctr);
  hipLaunchKernelGGL(spec_tree_emit_kernel, dim3(64), dim3(256), 0, ctx->stream, t, ord, upper, force_mask, walk, sparent, slevel,
                     info);
}

#ifdef PCV_EXPERIMENTS
size_t pcv_sample_count_scratch_words(uint32_t capacity, uint32_t max_open, int nlevels) {
  const size_t groups = (size_t)(nlevels + 2) / 3;
  return groups * (size_t)max_open * (kSlotCounters + 512u) + 2 * (size_t)capacity;
}
void pcv_launch_sample_tree_counts(pcv_ctx* ctx, const PcvNodeTableDev& t, const uint64_t* keys, uint32_t n, const PcvLevels& lv,
                                   double resolution, uint32_t max_points_per_node, uint32_t force_split_level1_mask, uint32_t* scratch,
                                   uint32_t saturate_above) {
  hipStream_t s = ctx->stream;
  const int groups = (lv.nlevels + 2) / 3;
  SampleCountTables c;
  c.max_open = t.max_open;
  c.counts = scratch;
  c.maps = c.counts + (size_t)groups * c.max_open * kSlotCounters;
  c.nslot = c.maps + (size_t)groups * c.max_open * 512u;
  c.nq = c.nslot + t.capacity;
  hipLaunchKernelGGL(sample_tree_init_kernel, dim3(1), dim3(256), 0, s, t, n, c);
  const unsigned grid = (unsigned)std::min<uint64_t>(512, ((uint64_t)n + 1023) / 1024 + 1);
  int cur = 0;
  for (int g = 0; g < groups; ++g) {
    {
      PcvProf prof(ctx, PCV_K_SPLIT_SEARCH);
      hipLaunchKernelGGL(sample_count_kernel, dim3(grid), dim3(256), 0, s, keys, n, g, groups, saturate_above, c);
    }
    {
      PcvProf prof(ctx, PCV_K_SPLIT_ASSIGN);
      hipLaunchKernelGGL(sample_tree_kernel, dim3(1), dim3(1024), 0, s, t, lv, resolution, max_points_per_node, g, cur, force_split_level1_mask, c);
    }
    const int levels_here = std::min(3, lv.nlevels - 3 * g);
    if (levels_here & 1) cur ^= 1;  // the open list flips once per level
  }
}

#endif  // PCV_EXPERIMENTS

void pcv_launch_pack_node_table(pcv_ctx* ctx, const PcvNodeTableDev& t, void* packed) {
  hipLaunchKernelGGL(pack_node_table_kernel, dim3(64), dim3(256), 0, ctx->stream, t, (uint8_t*)packed);
}

void pcv_launch_node_split(pcv_ctx* ctx, const PcvNodeTableDev& t, const void* sorted_keys, bool keys32, uint32_t n,
                           const PcvLevels& lv, double resolution, uint32_t max_points_per_node,
                           uint32_t force_split_level1_mask, const uint64_t* sorted_lo) {
  hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(init_root_kernel, dim3(1), dim3(256), 0, s, t, n);
  int k = 1;
  // two levels per launch pair (u64 keys of one word; the scratch holds the lists of every open node a level can have:
  // an open node holds more than `max_points_per_node` keys, + the root and the forced level-1 nodes)
  static const bool two_levels = [] {
    const char* e = pcv_experiment("PCV_SPLIT2");  // experiments: 0 = one level per launch pair
    return !e || atoi(e) != 0;
  }();
  if (two_levels && !keys32 && !sorted_lo && max_points_per_node > 0 && t.max_open >= (uint64_t)n / max_points_per_node + 16) {
    const unsigned grid = (unsigned)std::min<uint64_t>(t.max_open, (uint64_t)n / max_points_per_node + 16);
    int cur = 0;
    while (k <= lv.nlevels && k <= PCV_MAX_KEY_LEVELS) {
      const int second = (k + 1 <= lv.nlevels && k + 1 <= PCV_MAX_KEY_LEVELS) ? 1 : 0;
      {
        PcvProf prof(ctx, PCV_K_SPLIT_SEARCH);
        hipLaunchKernelGGL(split2_search_kernel<uint64_t>, dim3(grid), dim3(512), 0, s, t, (const uint64_t*)sorted_keys, k, cur,
                           max_points_per_node, force_split_level1_mask, second);
      }
      {
        PcvProf prof(ctx, PCV_K_SPLIT_ASSIGN);
        hipLaunchKernelGGL(split2_assign_kernel, dim3(1), dim3(1024), 0, s, t, lv, resolution, max_points_per_node, k, cur,
                           force_split_level1_mask, second);
      }
      k += 1 + second;
      cur ^= 1;
    }
  }
  for (; k <= lv.nlevels; ++k) {
    {
      PcvProf prof(ctx, PCV_K_SPLIT_SEARCH);
      if (keys32)
        hipLaunchKernelGGL(split_search_kernel<uint32_t>, dim3(512), dim3(256), 0, s, t, (const uint32_t*)sorted_keys, k,
                           sorted_lo);
      else
        hipLaunchKernelGGL(split_search_kernel<uint64_t>, dim3(512), dim3(256), 0, s, t, (const uint64_t*)sorted_keys, k,
                           sorted_lo);
    }
    {
      PcvProf prof(ctx, PCV_K_SPLIT_ASSIGN);
      hipLaunchKernelGGL(split_assign_kernel, dim3(1), dim3(1024), 0, s, t, lv, resolution, max_points_per_node, k,
                         force_split_level1_mask);
    }
  }
}
