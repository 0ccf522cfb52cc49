// pcv_internal.h — shared host-side declarations of the MI355X octree-build library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pcv_hip.h"
#include "pcv_spec.h"

#define PCV_HIP_CHECK(ctx, expr)                                                                   \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return (ctx)->fail(PCV_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// Per-level constants handed to every kernel by value (lands in SGPRs; uniform across the grid).
// edge[k], enc[k] for k = 0..nlevels: see pcv_level_table / reference codec.rs:31-40, node.rs:161.
// A path key word holds PCV_MAX_KEY_LEVELS (21) levels. Trees that need more (heavy duplicates in a cube with
// edge / resolution > 2^21) take the "deep" path: a second key word for levels 22..PCV_MAX_LEVELS. 40 levels is what the
// reference's NodeId can name (u128: 8 bits of level + 120 bits of index, node.rs:101-111).
#define PCV_MAX_LEVELS 40
struct PcvLevels {
  double root_min[3];
  double edge[PCV_MAX_LEVELS + 2];
  double inv_edge[PCV_MAX_LEVELS + 2];     // yh = RN(1 / edge[k]) for the exact constant-divisor division
  double inv_edge_lo[PCV_MAX_LEVELS + 2];  // yl = RN(1 / edge[k] - yh): the reciprocal as a double-double
  uint32_t enc[PCV_MAX_LEVELS + 3];  // 32-bit entries: a wave-uniform lv.enc[L] is a scalar load (a byte would be a vector load)
  // Octant digit of level k + 1 straight from the integer codes of level k (pcv_chain_dev.h, pcv_digit_from_codes):
  // digit_half[k] = 127 / 32767 when level k is u8 / u16-coded and the rounding-error bound holds there, else -1
  double digit_half[PCV_MAX_LEVELS + 2];
  // how the single chain pass gets the digit of level k + 1 (pcv_chain_dev.h): 0 = comparison against the cube centre,
  // 1 = from the integer codes of level k (digit_half[k] = 127 / 32767), 2 = from the Float32 codes of level k
  // (digit_half[k] = 0.5; a code of exactly 0.5 falls back to the comparison) — an integer so that the test is scalar
  uint32_t digit_mode[PCV_MAX_LEVELS + 2];
  // Encodings narrow with depth (the edge halves per level): levels [first_u16, first_u8) are u16-coded, levels from
  // first_u8 on u8-coded; both are "never" (a huge level) when the table is not monotone. The single chain pass runs one
  // straight-line loop per range instead of a switch per level.
  int32_t first_u16, first_u8;
  int32_t first_f32;  // levels [first_f32, first_u16) are Float32-coded ("never" when the table is not monotone)
  // Float32 codes of level k + 1 straight from the Float32 codes of level k (round 5; pcv_chain_dev.h "codes from codes"):
  // for the level steps k in [code_begin, code_end) the chain pass computes w = 2 v - bit per coordinate and keeps it as the
  // level-(k + 1) code wherever code_thr_hi[k] <= hi32(w) < hi32(1.0) for all three coordinates (code_thr_hi[k] = the high
  // word of the power of two below which a code of level k + 1 is too close to the rounding noise of the f64 chain to be
  // predicted: the wave then runs the full step). code_begin == code_end: no step admitted.
  uint32_t code_thr_hi[PCV_MAX_KEY_LEVELS + 2];
  int32_t code_begin, code_end;
  int32_t nlevels;  // number of digit levels materialised in the keys (<= PCV_MAX_KEY_LEVELS; <= PCV_MAX_LEVELS deep)
  int32_t fast_ok;  // root min and all edges are tame: unguarded exact division is valid for tame points
};

// A few persistent host threads for memory-bound host work next to the GPU (staging copies into pinned memory): a
// parallel-for over [0, count) that returns when every index is done. One per context, created on first use.
struct PcvHostPool {
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable wake, done;
  std::function<void(size_t)> job;
  size_t next = 0, count = 0, finished = 0;
  uint64_t generation = 0;
  bool stop = false;
  void start(unsigned n);
  void run(size_t n, const std::function<void(size_t)>& fn);
  ~PcvHostPool();
};

// Caching device allocator + pinned host scratch, one per context. Steady-state builds allocate nothing.
// A pool block assembled from separately created physical chunks mapped into one address range in a scrambled order
// (HIP virtual memory management; PcvPool::alloc explains why).
struct PcvVmmBlock {
  size_t size = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};
struct PcvPool {
  std::multimap<size_t, void*> free_blocks;
  std::map<void*, size_t> live;
  std::map<void*, PcvVmmBlock> vmm;  // blocks that must be unmapped / released instead of hipFree'd
  int device = 0;
  void free_block(void* p);
  void* alloc(size_t bytes, hipError_t* err);
  void release(void* p);
  void trim();
};

// Kernel ids for the optional per-launch HIP-event profile (pcv_ctx_set_profiling).
enum PcvKernelId {
  PCV_K_AABB = 0,
  PCV_K_CHAIN_KEYS,
  PCV_K_SORT_UPSWEEP64,
  PCV_K_SORT_SCAN,
  PCV_K_SORT_DOWNSWEEP64,
  PCV_K_SPLIT_SEARCH,
  PCV_K_SPLIT_ASSIGN,
  PCV_K_LEAF_ENCODE,
  PCV_K_SORT_UPSWEEP32,
  PCV_K_SORT_DOWNSWEEP32,
  PCV_K_PROMOTE_ENCODE,
  PCV_K_SORT_DOWNSWEEP_REC,
  PCV_K_CULL_NODES,
  PCV_K_VISIBLE_NODES,
  PCV_K_NODES_IN_LOCATION,
  PCV_K_CULL_POINTS,
  PCV_K_TRANSFORM_POINTS,
  PCV_K_QUERY_COMPACT,
  PCV_K_ROUTE_BUCKET,
  PCV_K_PARTITION_COUNT,
  PCV_K_PARTITION_SCATTER,
  PCV_K_PROMOTE_CLIMB,
  PCV_K_SPEC_ENCODE,
  PCV_K_RANK_HIST,
  PCV_K_SPEC_CONTINUE,
  PCV_K_SPEC_REPLAY,
  PCV_K_SORT_UPSWEEP_MAP,
  PCV_K_SORT_HIST_ROWS,
  PCV_K_CULL_NODES_SPARSE,
  PCV_K_SORT_SETTLE,  // the record sort's second pass settling the leaves' points itself (PcvSortFuse)
  PCV_K_INGEST,       // pcv_ingest_append: AoS -> SoA transposition + attribute copies + bounding-box fold of one batch
  PCV_K_COUNT
};

struct pcv_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string last_error;
  PcvPool pool;
  // small pinned mailbox for scalar read-backs (counters, flags): a D2H copy into pageable memory (a stack variable)
  // makes the runtime pin pages on the fly, which now and then costs milliseconds in the middle of a build
  uint64_t* mailbox = nullptr;  // 64 x u64 for read-backs + 64 x u64 reserved for the replay ranges of a build in flight
  uint64_t* mailbox_dev = nullptr;  // the same block as the device addresses it (kernels may store small verdicts there)
  // pinned host staging, grown on demand
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // second pinned block: staging of the single-chain build (sample table down, walk records / rank map up), so that the
  // node table can be staged in `pinned` while those uploads are still in flight
  void* pinned_spec = nullptr;
  size_t pinned_spec_bytes = 0;
  int pinned_spec_reserve(size_t bytes);
  // host -> device staging of pageable caller memory: ring of pinned chunks filled by the host pool, DMA per chunk
  static constexpr size_t kRingChunk = 32u << 20;
  static constexpr int kRingSlots = 3;
  void* ring[kRingSlots] = {};
  hipEvent_t ring_ev[kRingSlots] = {};
  bool ring_busy[kRingSlots] = {};
  int ring_next = 0;
  int ring_held = -1;  // a chunk an ingest is still filling (pcv_ingest.hip): the other users of the ring pass it over
  int ring_take() {
    int slot = ring_next;
    if (slot == ring_held) slot = (slot + 1) % kRingSlots;
    ring_next = (slot + 1) % kRingSlots;
    return slot;
  }
  PcvHostPool host_pool;
  int ring_ensure();
  int h2d(void* dst, const void* src, size_t bytes);  // asynchronous on `stream` from the device's point of view
  int h2d_fill(void* dst, size_t bytes, const std::function<bool(uint8_t* to, size_t off, size_t len)>& fill);
  hipEvent_t ev[PCV_NUM_STAGES + 2] = {};
  // per-stage begin / end events of the build in flight (a stage may be recorded out of order or not at all)
  hipEvent_t stage_b[PCV_NUM_STAGES] = {}, stage_e[PCV_NUM_STAGES] = {};
  bool stage_on[PCV_NUM_STAGES] = {}, stage_open[PCV_NUM_STAGES] = {};
  bool stage_times = false;  // PCV_BUILD_STAGE_TIMES of the build in flight: without it no stage event is recorded
  void stage_begin(int s) {
    if (!stage_times) return;
    (void)hipEventRecord(stage_b[s], stream);
    stage_on[s] = false;
    stage_open[s] = true;
  }
  void stage_end(int s) {  // a stage that was never begun in this build stays unmeasured
    if (!stage_open[s]) return;
    (void)hipEventRecord(stage_e[s], stream);
    stage_on[s] = true;
    stage_open[s] = false;
  }
  hipEvent_t xev = nullptr;  // stream hand-off with the caller's runtime (pcv_ctx_wait_stream / _signal_stream)
  hipEvent_t spec_ev = nullptr;  // single-chain build: "the predicted tree has reached the host"
  // side stream for small copies that must not sit between two kernels of `stream` (the predicted tree going down
  // while the chain pass starts; the node tables going up while the record sort runs). fork: side waits for what
  // `stream` has queued so far; join: `stream` waits for what the side stream has queued so far.
  hipStream_t side = nullptr;
  hipEvent_t side_fork = nullptr, side_join = nullptr;
  int side_begin() {
    if (hipEventRecord(side_fork, stream) != hipSuccess || hipStreamWaitEvent(side, side_fork, 0) != hipSuccess) return PCV_E_HIP;
    return PCV_OK;
  }
  int side_end() {
    if (hipEventRecord(side_join, side) != hipSuccess || hipStreamWaitEvent(stream, side_join, 0) != hipSuccess) return PCV_E_HIP;
    return PCV_OK;
  }
  // node / leaf records of K6: a context-owned device block (never recycled through the pool, so the side stream can
  // fill it while `stream` still runs kernels that use pool memory)
  void* table_dev = nullptr;
  size_t table_dev_bytes = 0;
  int table_dev_reserve(size_t bytes);

  // per-launch profile: event pairs recorded on `stream`, resolved after the next stream sync. 0 = off, 1 = every
  // launch, 2 = the kernels that move the whole cloud only (pcv_prof_is_major): an event pair costs the stream a few
  // microseconds, and the ~70 tiny launches of the sample phase are better read off the stage events
  int profiling = 0;
  struct ProfPending {
    int id;
    hipEvent_t a, b;
  };
  std::vector<ProfPending> prof_pending;
  std::vector<hipEvent_t> prof_free;
  uint64_t prof_launches[PCV_K_COUNT] = {};
  double prof_ms[PCV_K_COUNT] = {};
  hipEvent_t prof_event();
  void prof_resolve();

  int fail(int code, const std::string& msg) {
    last_error = msg;
    return code;
  }
  // cached pinned host blocks for result blobs (page-locking GBs costs more than the copy itself)
  std::multimap<size_t, void*> host_free;
  std::map<void*, size_t> host_live;
  int host_alloc(void** p, size_t bytes);
  void host_release(void* p);
  int dev_alloc(void** p, size_t bytes);
  void dev_free(void* p);
  int pinned_reserve(size_t bytes);
};

// Kernels whose launches touch every point of the cloud (profiling level 2 brackets only these).
inline bool pcv_prof_is_major(int id) {
  switch (id) {
    case PCV_K_AABB:
    case PCV_K_LEAF_ENCODE:
    case PCV_K_PROMOTE_ENCODE:
    case PCV_K_SORT_DOWNSWEEP_REC:
    case PCV_K_SORT_SETTLE:
    case PCV_K_PROMOTE_CLIMB:
    case PCV_K_SPEC_ENCODE:
    case PCV_K_RANK_HIST:
    case PCV_K_SORT_UPSWEEP_MAP:
    case PCV_K_SORT_UPSWEEP32:
    case PCV_K_SPEC_CONTINUE:
    case PCV_K_ROUTE_BUCKET:
    case PCV_K_PARTITION_COUNT:
    case PCV_K_PARTITION_SCATTER:
      return true;
    default:
      return false;
  }
}
// Brackets one kernel launch with HIP events on the ctx stream when profiling is on.
struct PcvProf {
  pcv_ctx* ctx;
  int id;
  hipEvent_t a = nullptr;
  PcvProf(pcv_ctx* c, int kernel_id) : ctx(c), id(kernel_id) {
    if (ctx->profiling == 1 || (ctx->profiling == 2 && pcv_prof_is_major(kernel_id))) {
      a = ctx->prof_event();
      (void)hipEventRecord(a, ctx->stream);
    }
  }
  ~PcvProf() {
    if (a) {
      hipEvent_t b = ctx->prof_event();
      (void)hipEventRecord(b, ctx->stream);
      ctx->prof_pending.push_back({id, a, b});
    }
  }
};

// RAII bundle of pool allocations released on scope exit (unless detached).
struct PcvScratch {
  pcv_ctx* ctx;
  std::vector<void*> ptrs;
  explicit PcvScratch(pcv_ctx* c) : ctx(c) {}
  ~PcvScratch() {
    for (void* p : ptrs) ctx->dev_free(p);
  }
  template <typename T>
  int get(T** out, size_t count) {
    void* p = nullptr;
    int rc = ctx->dev_alloc(&p, count * sizeof(T));
    if (rc != PCV_OK) return rc;
    ptrs.push_back(p);
    *out = (T*)p;
    return PCV_OK;
  }
  void detach(void* p) {
    for (auto& q : ptrs)
      if (q == p) {
        q = ptrs.back();
        ptrs.pop_back();
        return;
      }
  }
};

// ---- kernels (launchers; all asynchronous on `stream`) --------------------------------------------
// pcv_chain.hip
int pcv_launch_aabb(pcv_ctx* ctx, uint64_t n, const double* x, const double* y, const double* z, double* partial,
                    double* out6 /* device: min xyz, max xyz */);
// Routed input (multi-GPU build): instead of raw coordinates a point arrives as its level-1 chain state — the root
// octant digit and the level-1 codes (Float32 bit patterns; the exchange falls back to raw f64 when level 1 is not
// Float32-encoded). The chain continues at level 2 from decode(code) exactly as it would have on the sending rank.
struct PcvRouted {
  const uint8_t* oct = nullptr;  // null: raw points; the digit of point i is oct[i * oct_stride]
  uint32_t oct_stride = 1;
  const uint32_t* cx = nullptr;
  const uint32_t* cy = nullptr;
  const uint32_t* cz = nullptr;
};
// keys32: store the first 10 levels only as u32 (key >> 33); stride > 1: strided sample of the input.
// clump_shift: the strided sample is taken in clumps of 2^clump_shift consecutive points (0 = single points).
void pcv_launch_chain_keys(pcv_ctx* ctx, const PcvLevels& lv, uint64_t n, uint64_t stride, const double* x,
                           const double* y, const double* z, void* keys, bool keys32, const PcvRouted& routed = PcvRouted(),
                           uint32_t clump_shift = 0, uint32_t* zero = nullptr /* zero_words words (rounded up to 4) cleared on the way */,
                           size_t zero_words = 0);
void pcv_launch_depth_probe(pcv_ctx* ctx, const uint64_t* sorted, uint32_t n, uint32_t gap, uint32_t* out);

// pcv_sort.hip — stable LSD radix sort, 8-bit digits, reduce-then-scan with LDS histograms.
struct PcvSortPayload {
  void* vec_in = nullptr;   // optional 16-byte payload word per key (uint4), ping-pong partner in vec_out
  void* vec_out = nullptr;
  int vec_bytes = 16;       // 8: 12-byte records (uint2 payload, key = rank << 8 | blue), single-chain build only
  int nwords = 0;           // extra 32-bit payload planes that travel with the key (0..8)
  uint32_t* in[8] = {};
  uint32_t* out[8] = {};
  // set: the FIRST pass reads plane 0 from here instead of in[0] (the caller's own array in record order — the intensity
  // plane of the single-chain build needs no copy into the sort's buffers); later passes ping-pong between out[0] and in[0]
  const uint32_t* first_in0 = nullptr;
  // set (round 6): the records arrive WITHOUT their colour — key = rank << 8, the upper half of payload .y empty — and the first
  // pass of the 12-byte record sort reads r, g, b of record i at color_in + i * color_stride (the caller's array, input order) as
  // it loads the record; what it writes is the record with colour (blue in the key's low byte, red / green in payload .y)
  const uint8_t* color_in = nullptr;
  uint32_t color_stride = 3;
};
size_t pcv_sort_scratch_bytes(uint64_t n);
bool pcv_sort_first_pass_joins_color(uint64_t n);
// Sorts keys_in -> ... ping-pong between (keys_a, payload.in) and (keys_b, payload.out). Returns in
// *result_in_a whether the final sorted data is in the a-side (true) or b-side (false).
int pcv_radix_sort_u64(pcv_ctx* ctx, uint64_t* keys_a, uint64_t* keys_b, uint64_t n, int begin_bit, int end_bit,
                       PcvSortPayload* payload, void* scratch, bool* result_in_a);
int pcv_radix_sort_u32(pcv_ctx* ctx, uint32_t* keys_a, uint32_t* keys_b, uint64_t n, int begin_bit, int end_bit,
                       PcvSortPayload* payload, void* scratch, bool* result_in_a);
#ifdef PCV_EXPERIMENTS
// The sample's key sort, one launch per 9-bit digit (pcv_sort.hip: onesweep_keys_kernel). `scratch`: pcv_onesweep_scratch_words
// words whose first pcv_onesweep_zero_words are zero when the sort starts.
bool pcv_onesweep_fits(uint64_t n, int bits);
size_t pcv_onesweep_zero_words(uint64_t n, int bits);
size_t pcv_onesweep_scratch_words(uint64_t n, int bits);
int pcv_sort_keys_onesweep(pcv_ctx* ctx, uint64_t* keys_a, uint64_t* keys_b, uint64_t n, int begin_bit, int end_bit, uint32_t* scratch,
                           bool* result_in_a, int diag = 0 /* timing-only variants (pcv_exp_time_key_sort, libpcv_hip_exp.so) */);
#endif

// rows (pcv_launch_rank_hist_rows, map_entries counters per sort workgroup): the first pass takes its histogram from them and
// applies the map inside its downsweep — the keys are not read an extra time
// `second` set: a sort that takes the two-pass form with both histograms from the rank counts queues its first pass (and the
// second one's layout) only and leaves what the second pass needs there (second->pending; *result_in_a already says where
// the second pass will put the records); pcv_radix_sort_records_second queues that pass — plain, or settling the leaves' points
// itself (PcvSortFuse). Any other form of the sort runs to its end and leaves second->pending false.
struct PcvSortSecond {
  bool pending = false;
  bool join_side = false;          // the pass's layout kernels were queued on the context's side stream
  const uint32_t* src = nullptr;   // keys after the first pass (rank << 8 | blue)
  uint32_t* dst = nullptr;
  const void* vec_src = nullptr;   // uint2 payloads after the first pass
  void* vec_dst = nullptr;
  const uint32_t* plane_src = nullptr;  // the intensity plane travelling with the records, or null
  uint32_t* plane_dst = nullptr;
  uint64_t n = 0, chunk = 0;
  int pieces = 0, shift = 0, nbits = 0;
  int low_bits = 0, blocks = 1;  // piece k holds the records whose rank's lower `low_bits` bits are k / blocks
  const uint32_t *hist = nullptr, *totals = nullptr, *order = nullptr;
  const void* ranges = nullptr;
};
struct PcvNodeRec;
// The second pass of the record sort as the producer of the final bytes: every run it would write is one leaf's records at
// consecutive sorted slots (a piece holds ONE value of the rank's lower digit, the pass's digit is the upper one), so for the
// leaves flagged here it does `settle`'s work on the record in flight — final rewrite + stores for seven of eight, the 16-byte
// climber record for every eighth — instead of writing 12 bytes that `settle` would read back (24 bytes per point less).
struct PcvSortFuse {
  const PcvNodeRec* leaf_rec = nullptr;  // per true leaf rank
  const uint8_t* leaf_fused = nullptr;   // per true leaf rank: 1 = settled here (u8 / u16 codes, no continuation, no replay, not the root)
  const uint32_t* climb_base = nullptr;
  void* climbers = nullptr;              // uint4 per climber; PcvClimber (32 bytes) with an intensity plane
  uint8_t* xyz_blob = nullptr;
  uint8_t* rgb_blob = nullptr;
  uint8_t* inten_blob = nullptr;         // set when the records travel with an intensity plane
  uint32_t num_leaves = 0;
  uint32_t low_bits = 0, blocks = 1;  // (filled by pcv_radix_sort_records_second from the held-back pass)
};
int pcv_radix_sort_records_mapped(pcv_ctx* ctx, uint32_t* keys_a, uint32_t* keys_b, uint64_t n, int key_bits,
                                  PcvSortPayload* payload, void* scratch, const uint32_t* map, uint32_t map_entries,
                                  bool* result_in_a, const uint32_t* rows = nullptr, PcvSortSecond* second = nullptr);
int pcv_radix_sort_records_second(pcv_ctx* ctx, PcvSortSecond* second, const PcvSortFuse* fuse /* or null: a plain pass */);

// pcv_topology.hip — node split (topology from sorted keys).
// Device node table, structure of arrays, BFS order (level-major, prefix-sorted inside a level).
struct PcvNodeTableDev {
  uint32_t capacity;
  uint64_t* prefix;      // left-aligned path key (digits beyond `level` are zero)
  uint64_t* prefix_lo;   // deep trees: digits of levels 22.. (level k at bits 3 * (42 - k)); null otherwise
  uint32_t* lo;          // [lo, hi) range in the sorted key array
  uint32_t* hi;
  uint32_t* parent;
  uint32_t* first_child; // index of the first child (children are contiguous, in digit order)
  uint8_t* level;
  uint8_t* child_mask;   // bit c set = child c exists
  uint8_t* open;         // 1 = split further (inner node), 0 = leaf
  uint32_t* bounds;      // scratch: 9 bounds per node of the level being expanded (capacity x 9), or the lists and bounds of
                         // the two-levels-per-launch kernels (85 x max_open), whichever is larger
  uint32_t max_open = 0; // open nodes a level can have as far as the scratch goes (0: one level per launch pair only)
  uint32_t* counters;    // [0] node_count, [1] error flag, [2..] level_start[k] (k = 0..PCV_MAX_LEVELS+1)
};
// One node of the device table, packed for a single device-to-host copy (pcv_launch_pack_node_table): the copy starts
// with the 64 counters (256 bytes), the nodes follow.
struct PcvPackedNode {
  uint64_t prefix;
  uint32_t lo, hi, first_child;
  uint8_t level, child_mask, open, pad;
};
static_assert(sizeof(PcvPackedNode) == 24, "packed node");
constexpr size_t kPcvPackHeader = 256;
void pcv_launch_pack_node_table(pcv_ctx* ctx, const PcvNodeTableDev& t, void* packed /* header + capacity nodes */);
// single-chain build: the sample's node table by counting the (unsorted) sample keys, three levels per launch pair
// (pcv_topology.hip); scratch: pcv_sample_count_scratch_words() u32. Needs t.max_open > 0 and one-word keys.
size_t pcv_sample_count_scratch_words(uint32_t capacity, uint32_t max_open, int nlevels);
void pcv_launch_sample_tree_counts(pcv_ctx* ctx, const PcvNodeTableDev& t, const uint64_t* keys, uint32_t n, const PcvLevels& lv,
                                   double resolution, uint32_t max_points_per_node, uint32_t force_split_level1_mask, uint32_t* scratch,
                                   uint32_t saturate_above /* counters may stop growing beyond this: above the split threshold AND the candidate band */);
// single-chain build: predicted tree on the device from the sample's node table. ord: capacity u32 of scratch; walk /
// sparent: 1 + 8 x capacity u32; slevel: 1 + 8 x capacity bytes; info: 4 u32 (nodes, split error flags, sample nodes,
// any candidate)
void pcv_launch_spec_tree(pcv_ctx* ctx, const PcvNodeTableDev& t, double upper, uint32_t force_mask, uint32_t* ord, uint32_t* walk,
                          uint32_t* sparent, uint8_t* slevel, uint32_t* info, uint32_t* pool_ctr /* kPcvPoolRegions counters, zeroed here; may be null */);
// single-chain build: the predicted-leaf -> true-leaf rank map on the device, from the exact counts (`counts`: one u32 per
// T'' node, leaf entries filled by pcv_launch_rank_hist, inner entries zero). tn = number of T'' nodes (the host knows it
// from its mirror of the tree); nst / base: tn u32 of scratch each; out: [0] number of true leaves, [1] 1 = prediction too
// shallow
void pcv_launch_spec_resolve(pcv_ctx* ctx, const PcvLevels& lv, double resolution, uint32_t cap, uint32_t force_mask, const uint32_t* walk,
                             const uint8_t* slevel, uint32_t tn, uint32_t* counts, uint32_t* nst, uint32_t* base, uint32_t* map,
                             uint32_t* out);
// sorted_lo (deep trees): second key word, sorted together with the first; levels > PCV_MAX_KEY_LEVELS search it
void pcv_launch_node_split(pcv_ctx* ctx, const PcvNodeTableDev& t, const void* sorted_keys, bool keys32, uint32_t n,
                           const PcvLevels& lv, double resolution, uint32_t max_points_per_node,
                           uint32_t force_split_level1_mask, const uint64_t* sorted_lo = nullptr);
// deep trees: digits of up to PCV_MAX_LEVELS levels as four 32-bit words (hi >> 32, hi, lo >> 32, lo); hi holds levels
// 1..21 as in the ordinary key, lo holds level k > 21 at bits 3 * (42 - k)
void pcv_launch_chain_keys_deep(pcv_ctx* ctx, const PcvLevels& lv, uint64_t n, const double* x, const double* y,
                                const double* z, const PcvRouted& routed, uint32_t* const words[4]);
void pcv_launch_combine_words(pcv_ctx* ctx, uint64_t n, const uint32_t* const words[4], uint64_t* hi, uint64_t* lo);

// pcv_encode.hip — leaf lookup + leaf-level encode (input order), promotion + final encode (sorted order).
struct PcvWalkTables {
  uint32_t num_nodes;
  const uint64_t* walk;  // per node: first_child(32) | child_mask(8) << 32 | leaf(1) << 40 | level(8) << 48;
                         // for leaves the low 32 bits hold the leaf rank
};
// record = rank (u32) + payload uint4 {code x, code y, code z, rgba} [+ planes: intensity bits, Float64 high words]
void pcv_launch_leaf_encode(pcv_ctx* ctx, const PcvLevels& lv, const PcvWalkTables& wt, uint64_t n, const double* x,
                            const double* y, const double* z, const PcvRouted& routed, const uint8_t* color,
                            uint32_t color_stride, const float* intensity, uint32_t* rank, void* payload /* uint4[n] */,
                            uint32_t* cx_hi,
                            uint32_t* cy_hi, uint32_t* cz_hi, uint32_t* inten_bits);

// single-chain build (pcv_spec.h): the one chain pass down the predicted tree, the exact per-leaf counts, and the
// rank / payload fix-up once the true tree is known.
// 12-byte records (`wide` set, u32 + uint2 per point instead of u32 + uint4): key = rank << 8 | blue, payload =
// {code x | code y << 16, code z | red << 16 | green << 24} for u8 / u16-coded leaf levels; a point whose leaf level is
// Float32-coded names an entry of the `wide` POOL in the first payload word and keeps its three 32-bit codes there — 24
// instead of 40 bytes per point and pass through the record sort. The pool has kPcvPoolRegions regions of
// pcv_pool_region_entries(n) entries; the points of input slice s (1 024 consecutive points) use region s % kPcvPoolRegions,
// filled from its first entry upwards by ONE reservation per wave on the region's counter (a single counter for the whole
// pool serialises: 7.3 instead of 2.2 ms for the pass at 100 M points) — so the used part of the pool is kPcvPoolRegions
// dense prefixes (100 MB for the 6.3 M entries of the bench cloud); the rare replay takes entries from the regions' tops.
void pcv_launch_spec_encode(pcv_ctx* ctx, const PcvLevels& lv, const uint32_t* walk, uint64_t n, const double* x,
                            const double* y, const double* z, const PcvRouted& routed, const uint8_t* color,
                            uint32_t color_stride, const float* intensity, uint32_t* rank, void* payload /* uint4[n] */,
                            uint32_t* inten_bits, uint8_t* depth_grid, void* wide, uint32_t* pool_ctr, const uint32_t* tree_info,
                            bool color_late = false /* 12-byte records leave without their colour (PcvSortPayload::color_in) */,
                            uint32_t* zero = nullptr /* with depth_grid: zero_words words (16-byte aligned, a multiple of 4) cleared before the pass */,
                            size_t zero_words = 0);
// The colour joined into records that left the chain pass without it, in place (the record sort's first pass does this on the
// fly; this pass exists for the sorts that cannot: the rank-count rows did not fit, experiments).
void pcv_launch_join_color(pcv_ctx* ctx, uint64_t n, const uint8_t* color, uint32_t color_stride, uint32_t* keys, void* payload_uint2);
constexpr uint32_t kPcvPoolRegions = 1024;
// entries per region: every slice of 1 024 points could be all Float32-coded
inline uint64_t pcv_pool_region_entries(uint64_t n) { return (((n + 1023) / 1024 + kPcvPoolRegions - 1) / kPcvPoolRegions) * 1024; }
size_t pcv_spec_depth_grid_bytes();  // scratch for the depth-prediction grid of the binned pass
void pcv_launch_rank_hist(pcv_ctx* ctx, const uint32_t* rank, uint64_t n, uint32_t num_bins, uint32_t* counts /* zeroed */,
                          int shift = 0);
uint32_t pcv_rank_hist_max_bins();
void pcv_launch_rank_hist_rows(pcv_ctx* ctx, const uint32_t* rank, uint64_t n, uint32_t num_bins, uint32_t* counts /* zeroed */,
                               int shift, int groups, uint64_t chunk, uint32_t* rows /* groups x num_bins */);
// workgroups and keys per workgroup of the 12-byte record sort (tiles of 8 192)
void pcv_sort_rec12_geometry(uint64_t n, int* groups, uint64_t* chunk);
// Chain continuation of the true leaves below a split first candidate (pcv_spec.h): `ranges` = device array of
// pcv_cont_range_bytes()-sized records filled by pcv_fill_cont_range (levels + cube min of the candidate), `items` = one
// entry per <= kPcvSettleTile sorted slots of one such leaf (rank = index of its range); rewrites the codes in place.
size_t pcv_cont_range_bytes();
void pcv_fill_cont_range(void* dst, uint32_t from_level, uint32_t to_level, const double mn[3]);
void pcv_launch_spec_continue(pcv_ctx* ctx, const PcvLevels& lv, const void* ranges, const PcvSettleItem* items, uint32_t num_items,
                              void* sorted_payload, void* wide = nullptr);
// ranges: device array of {first sorted slot, flagged slots before it, level, pad} (4 x u32), after the record sort
void pcv_launch_spec_replay(pcv_ctx* ctx, const PcvLevels& lv, const void* ranges, uint32_t num_ranges, uint32_t total,
                            const double* x, const double* y, const double* z, const PcvRouted& routed, void* sorted_payload,
                            void* wide = nullptr, uint32_t pool_cap = 0 /* pcv_pool_region_entries(n) */);

// Everything K6 needs about a node in one 80-byte record, so a slot's dependent loads are rank -> record (-> the
// parent's record per climb) instead of chained table lookups (node, level, per-level edge/encoding, offsets).
struct alignas(16) PcvNodeRec {
  uint32_t lo;         // leaves: first sorted slot of the leaf
  uint32_t parent;     // node index of the parent, 0xffffffff for the root
  uint32_t child_off;  // offset of this node's promoted block inside the parent's stream
  uint32_t enc;        // PCV_ENC_* of the node's level
  uint64_t xyz_off;    // byte offset of the node's .xyz content in the xyz blob
  uint64_t point_off;  // point offset in the rgb / intensity blobs
  double mn[3];        // cube min (NodeId::find_bounding_cube recurrence)
  double edge;         // cube edge of the node's level
  double inv_edge;     // RN(1 / edge), 0 when the exact-division fast path must not be used
  double inv_edge_lo;  // RN(1 / edge - inv_edge)
};
struct PcvPromoteTables {
  const PcvNodeRec* leaf_rec;  // per leaf rank
  const PcvNodeRec* node_rec;  // per node index
};
// climb_base[leaf rank] = number of climbers (every 8th point of a non-root leaf) in the leaves before it; climbers:
// pcv_climber_bytes(num_climbers) bytes of scratch that `settle` fills and `climb` consumes
size_t pcv_climber_bytes(uint64_t num_climbers);
bool pcv_climb16_enabled();  // leaf-wise kernels without intensity / Float64 planes keep 16-byte climber records
void pcv_launch_promote_encode(pcv_ctx* ctx, const PcvLevels& lv, const PcvPromoteTables& pt, uint64_t n,
                               const uint32_t* rank, const void* payload /* uint4[n] */, const uint32_t* cx_hi,
                               const uint32_t* cy_hi, const uint32_t* cz_hi, const uint32_t* inten_bits,
                               const uint32_t* climb_base, uint32_t num_climbers, void* climbers, uint8_t* xyz_blob,
                               uint8_t* rgb_blob, uint8_t* inten_blob, const void* wide = nullptr,
                               const PcvSettleItem* items = nullptr, uint32_t num_items = 0,
                               const PcvSettleItem* climb_items = nullptr, uint32_t num_climb_items = 0,
                               const void* cont_ranges = nullptr /* leaf-wise settle: items with pad != 0 continue their chain
                                                                    from range pad - 1 (pcv_fill_cont_range) */);

struct PcvOctreeQuery;  // device-resident traversal tables (pcv_query.hip)

// The finished octree (node table + node-contiguous blobs).
struct PcvBuild;  // pcv_build.hip: state between pcv_build_begin and pcv_build_finish
struct pcv_octree {
  pcv_ctx* ctx = nullptr;
  PcvBuild* pending = nullptr;
  double resolution = 0;
  double bbox_min[3] = {0, 0, 0}, bbox_max[3] = {0, 0, 0};
  bool has_intensity = false;
  uint64_t num_points = 0;
  std::vector<pcv_node_info> nodes;
  uint8_t *d_xyz = nullptr, *d_rgb = nullptr, *d_int = nullptr;
  uint64_t xyz_bytes = 0, rgb_bytes = 0, int_bytes = 0;
  // host copies of the blobs, pinned (hipHostMalloc) so the D2H runs at link speed and nothing is zero-filled first
  struct HostBlob {
    uint8_t* p = nullptr;
    uint8_t* data() const { return p; }
  } h_xyz, h_rgb, h_int;
  bool host_valid = false;
  float stage_ms[PCV_NUM_STAGES] = {};
  int key_levels = 0;    // digit levels the key sort covered (depth speculation)
  int key_attempts = 0;  // 0 = single-chain build, 1 = depth speculation held (or was off), 2+ = redone
  int record_bytes = 0;  // bytes per record in the record sort (20, or 12 packed)
  uint64_t spec_stats[4] = {};  // single-chain build: nodes / leaves of the predicted tree, points in an unsplit first
                                // candidate (their kept codes are their leaf codes), points that replayed the chain
  uint64_t wide_pool_entries = 0;  // single-chain build, 12-byte records: entries of the Float32-code pool the chain pass used
  uint64_t settled_in_sort = 0;    // points of the leaves the record sort's second pass finished itself (PcvSortFuse)
  uint64_t spec_continued = 0;  // points whose chain was continued from the codes of a split candidate
  PcvOctreeQuery* query = nullptr;
  // octrees opened from a directory: node files are read on demand
  std::string directory;
  std::map<std::pair<uint64_t, int>, std::vector<uint8_t>> file_cache;
};
int pcv_octree_prepare_query(pcv_octree* t);
void pcv_octree_release_query(pcv_octree* t);
int pcv_octree_fetch_host(pcv_octree* t);
int pcv_octree_load_device(pcv_octree* t);  // directory octrees: read + upload all node files (pcv_io.cpp)
int pcv_octree_read_node_file(pcv_octree* t, uint64_t i, int which, const uint8_t** data, uint64_t* len);
int pcv_bytes_per_coordinate(uint32_t enc);

// host helpers (pcv_build.hip)
int pcv_make_levels(const double bmin[3], const double bmax[3], double resolution, int cap, PcvLevels* lv,
                    int* max_level, std::vector<double>* edges, std::vector<int32_t>* encs);
