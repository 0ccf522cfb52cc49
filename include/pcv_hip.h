/* pcv_hip.h — C ABI of the MI355X-native octree-build / cull hot path of point_cloud_viewer.
 *
 * The reference (Rust) has no FFI boundary for this path; the boundary is the crate's public
 * surface. Each entry point below names the reference item it replaces (paths relative to the
 * reference checkout). A Rust veneer (point_cloud_viewer_amd/rust_shim/) binds these
 * 1:1 and keeps `build_octree`, `Octree`, `NodeId`, `PointCulling` as the user-facing names.
 *
 * Rules of the ABI
 *  - plain pointers and sizes only; the caller owns every input buffer, the library never frees them;
 *  - every function returns an int status (PCV_OK or a negative PCV_E_*), never unwinds;
 *    pcv_last_error(ctx) holds a human-readable message for the last failure on that context;
 *  - a pcv_ctx is bound to one HIP device + one stream and is NOT thread-safe; use one per host
 *    thread/device (reference: the build uses the global rayon pool, src/bin/build_octree.rs:43-46);
 *  - buffers are tagged PCV_MEM_HOST or PCV_MEM_DEVICE; device buffers must live on the ctx's device.
 */
#ifndef PCV_HIP_H
#define PCV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): pcv_ingest_*, PCV_ROUTE_OCTANTS_ONLY, PCV_STAGE_SORT_SECOND (PCV_STAGE_TOTAL moved from 8 to 9) */
#define PCV_ABI_VERSION 2

/* status codes (reference: error-chain kinds src/errors.rs:18-48; the builder itself panics) */
#define PCV_OK 0
#define PCV_E_INVALID (-1) /* bad argument (ErrorKind::InvalidInput) */
#define PCV_E_HIP (-2)     /* HIP runtime failure */
#define PCV_E_IO (-3)      /* file system failure (ErrorKind::Io) */
#define PCV_E_OOM (-4)     /* device or host allocation failed / node table capacity exceeded */
#define PCV_E_DEPTH (-5)   /* a node at level 40 would still have to be split (the reference's NodeId ends there too) */
#define PCV_E_NOT_FOUND (-6) /* ErrorKind::NodeNotFound */

#define PCV_MEM_HOST 0
#define PCV_MEM_DEVICE 1

/* position encodings == proto PositionEncoding values (point_viewer_proto_rust/src/proto.proto:82-88) */
#define PCV_ENC_UINT8 1
#define PCV_ENC_UINT16 2
#define PCV_ENC_FLOAT32 3
#define PCV_ENC_FLOAT64 4

/* path digits kept per point: 3 bits per level in a 64-bit key word; deeper trees (up to 40 levels, all the
 * reference's u128 NodeId can name) use a second word inside the library */
#define PCV_MAX_KEY_LEVELS 21
/* reference src/octree/generation.rs:37 */
#define PCV_DEFAULT_MAX_POINTS_PER_NODE 100000u

typedef struct pcv_ctx pcv_ctx;
typedef struct pcv_octree pcv_octree;

/* ---- context -------------------------------------------------------------------------------- */
/* `stream` is a hipStream_t (may be NULL = a stream owned by the context). The context also owns a small side stream
 * for copies that would otherwise sit between two kernels (the build's host mirror of the predicted tree, the upload
 * of the node tables); everything queued there is joined back into `stream` before a kernel that depends on it, so
 * the caller only ever orders itself against `stream` (pcv_ctx_wait_stream / pcv_ctx_signal_stream). */
int pcv_ctx_create(int device, void* stream, pcv_ctx** out);
void pcv_ctx_destroy(pcv_ctx* ctx);
const char* pcv_last_error(const pcv_ctx* ctx);
int pcv_abi_version(void);
/* Wait for everything queued on the context's stream (the few entry points documented as asynchronous). */
int pcv_ctx_synchronize(pcv_ctx* ctx);
/* Stream hand-off with the caller's runtime (torch, RCCL, another library) without blocking the host:
 * pcv_ctx_wait_stream   - everything queued on the context's stream AFTER this call runs after what is queued on
 *                         `stream` now (the caller produced the inputs there);
 * pcv_ctx_signal_stream - what is queued on `stream` after this call runs after the context's work queued so far
 *                         (the caller consumes device-resident outputs there).
 * `stream` is a hipStream_t of the context's device; NULL names the legacy default stream. Because NULL in
 * pcv_ctx_create means "own stream", a caller whose work sits on the default stream MUST order it with these two
 * calls (or synchronise the device) before handing buffers over. */
int pcv_ctx_wait_stream(pcv_ctx* ctx, void* stream);
int pcv_ctx_signal_stream(pcv_ctx* ctx, void* stream);
/* Release cached device/host scratch held by the context. */
int pcv_ctx_trim(pcv_ctx* ctx);

/* Optional per-launch profile: enabled = 1 brackets every kernel launch of this context with HIP events on the
 * context's stream, enabled = 2 only the kernels that pass over the whole cloud (an event pair costs the stream a few
 * microseconds; a build makes ~100 small launches); pcv_ctx_kernel_stats returns, per kernel id (0 .. return value - 1),
 * the kernel's name, the number of launches and their summed duration since the last reset. */
int pcv_ctx_set_profiling(pcv_ctx* ctx, int enabled);
int pcv_ctx_reset_kernel_stats(pcv_ctx* ctx);
int pcv_ctx_kernel_stats(pcv_ctx* ctx, int kernel_id, const char** name, uint64_t* launches, double* total_ms);

/* ---- inputs --------------------------------------------------------------------------------- */
/* One batch of points, SoA. Replaces `PointsBatch` (src/lib.rs:102-107): positions Vec<Point3<f64>>,
 * "color" U8Vec3 and optional "intensity" F32 (src/octree/mod.rs:62-74). */
typedef struct pcv_points {
  uint64_t n;
  const double* x;
  const double* y;
  const double* z;
  const uint8_t* color;   /* n * color_stride bytes, r,g,b first; required */
  uint32_t color_stride;  /* 3 (as in .rgb files) or 4 (rgba, alpha ignored) */
  const float* intensity; /* NULL = no "intensity" attribute */
  int32_t mem;            /* PCV_MEM_HOST or PCV_MEM_DEVICE (all pointers alike) */
} pcv_points;

/* Arguments of `build_octree` (src/octree/generation.rs:289-295). */
typedef struct pcv_build_params {
  double resolution;            /* meters; reference CLI default 0.001 (src/bin/build_octree.rs:33) */
  double bbox_min[3];           /* `bounding_box: Aabb`; may be loose */
  double bbox_max[3];
  uint32_t max_points_per_node; /* 0 = PCV_DEFAULT_MAX_POINTS_PER_NODE (the reference's constant) */
  uint32_t flags;               /* PCV_BUILD_* */
} pcv_build_params;

/* Compute the bounding box on the device first (== build_octree_from_file's find_bounding_box pass,
 * generation.rs:256-287); bbox_min/max are then outputs. */
#define PCV_BUILD_COMPUTE_BBOX 1u
/* Exact two-chain pipeline with full-depth path keys: disables the single-chain build and the sampled depth
 * speculation (same result, slower). */
#define PCV_BUILD_NO_SPECULATION 2u
/* Take the single-chain build (topology predicted from a strided sample, ONE chain pass, exact per-leaf counts decide
 * the tree; see csrc/pcv_spec.h) whatever the input size; by default it is used from 2^22 points on. Same result: if
 * the prediction does not cover the tree the exact pipeline redoes the build. */
#define PCV_BUILD_FORCE_SINGLE_CHAIN 4u
/* Never take the single-chain build (the exact pipeline with its depth speculation runs instead). */
#define PCV_BUILD_NO_SINGLE_CHAIN 8u
/* Single-chain build, diagnostics: the device's predicted-leaf -> true-leaf rank map (which drives the record sort) is
 * downloaded and compared entry by entry with the host's (which the node tables come from) before the octree is handed
 * out; PCV_E_HIP on a difference. Without the flag only the number of true leaves and the "too shallow" verdict of the two
 * resolves are compared (always, for free). */
#define PCV_BUILD_CHECK_RESOLVE 16u
/* Record the GPU time of every stage of the build (pcv_octree_stage_ms). Off by default: the 16 event records of a build
 * cost its stream ~0.1 ms (measured: 5.31 -> 5.21 ms per 100 M-point build), which a caller who does not read the stage
 * times should not pay; without the flag only PCV_STAGE_TOTAL is measured and the other stages read 0. */
#define PCV_BUILD_STAGE_TIMES 32u

/* Multi-GPU build: level-1 nodes whose bit is set are split even if this rank's share of their points
 * is below the capacity — the split decision of the GLOBAL tree, made from the all-reduced bucket counts. */
#define PCV_BUILD_FORCE_SPLIT_L1(mask8) (((uint32_t)(mask8) & 0xffu) << 8)

/* ---- the build ------------------------------------------------------------------------------ */
/* Replaces build_octree (generation.rs:289-403) up to, but not including, the file writes:
 * the result holds the finished node table and node-contiguous .xyz/.rgb/.intensity bytes. */
int pcv_build_octree(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points, pcv_octree** out);

/* Limits of one call (stated here, not discovered at run time): positions are addressed with 32 bits, so a build takes at
 * most 2^32 - 2 points (PCV_E_INVALID above that); everything is device-resident, about 80 bytes of HBM per point at the
 * peak of a build (27-31 B of input + records, rank counts, the wide-code pool and the output blobs), so ONE 288 GB MI355X
 * ends at roughly 3 x 10^9 points. The reference streams any size through node files (generation.rs:58-126); larger clouds
 * go through the multi-GPU path (pcv_route_* + pcv_build_begin_routed), which shards by subtree. */
#define PCV_MAX_POINTS_PER_BUILD 0xfffffffeull

/* ---- streaming batch ingest: the reference's `impl Iterator<Item = PointsBatch>` (generation.rs:289-295) ------------
 * A PointsBatch (src/lib.rs:102-107) holds `position: Vec<Point3<f64>>` — AoS, 24 bytes per point — "color"
 * Vec<Vector3<u8>> (3 bytes per point) and optionally "intensity" Vec<f32> (src/octree/mod.rs:62-74), and arrives 500 000
 * points at a time (src/lib.rs:52). These calls take the batches AS THEY ARE, one at a time:
 *   pcv_ingest_begin   num_points_hint = NumberOfPoints::num_points() of the stream (0 = unknown: the device arrays grow);
 *                      has_intensity = the attribute list names "intensity" (then every batch must carry it).
 *   pcv_ingest_append  copies the batch's three arrays side by side into the chunk of the context's pinned ring it has in
 *                      hand; a chunk that cannot take the next batch goes up in ONE DMA (32 MiB: two batches of 500 000
 *                      points), followed by ONE kernel per batch that transposes the positions AoS -> SoA into place behind
 *                      the points already on the device, copies colour / intensity behind theirs and folds the batch into
 *                      the running bounding box (find_bounding_box, generation.rs:256-270). Returns when the batch is in
 *                      the chunk — the caller's arrays are free again and the next batch can be produced while earlier
 *                      ones go up. Batches of any size (split into pieces of 2^20 points inside); n == 0 is a no-op. Host
 *                      memory: the ring (3 x 32 MiB), whatever the size of the cloud.
 *   pcv_ingest_bbox    the bounding box of the points appended so far (waits for the queued batches); Aabb::zero() for none.
 *   pcv_ingest_finish  pcv_build_octree on the ingested cloud; with PCV_BUILD_COMPUTE_BBOX the box folded during the ingest
 *                      is used (no pass over the cloud), otherwise params->bbox_* as build_octree takes it from its caller.
 *                      ALWAYS consumes the ingest, whatever it returns.
 *   pcv_ingest_abort   drops an ingest without building.
 * Other calls on the context between begin and finish (another ingest, a build from host arrays) are allowed: the chunk in
 * hand is theirs to pass over, not to reuse. */
typedef struct pcv_ingest pcv_ingest;
int pcv_ingest_begin(pcv_ctx* ctx, uint64_t num_points_hint, int has_intensity, pcv_ingest** out);
int pcv_ingest_append(pcv_ingest* ingest, const double* xyz /* n x 3, x y z per point (host) */, const uint8_t* rgb /* n x 3 (host) */,
                      const float* intensity /* n (host), NULL without the attribute */, uint64_t n);
uint64_t pcv_ingest_num_points(const pcv_ingest* ingest);
int pcv_ingest_bbox(pcv_ingest* ingest, double bbox_min[3], double bbox_max[3]);
int pcv_ingest_finish(pcv_ingest* ingest, const pcv_build_params* params, pcv_octree** out);
void pcv_ingest_abort(pcv_ingest* ingest);

/* The same build in two steps, for the multi-GPU path: when the level-2 subtrees of one level-1 node live on
 * different ranks, the every-8th promotion (generation.rs:195-253) into the level-1 node and into the root runs over
 * streams that span ranks, so the stream offsets must be agreed between the topology and the encode phase.
 *   pcv_build_begin       K1..K4 + the bottom-up stream lengths |pre(node)| of the local tree
 *   pcv_build_top_streams lengths of the local level-1 / level-2 streams (0 = node absent on this rank)
 *   pcv_build_finish      K5, record sort, K6. With a layout, the root and the level-1 nodes are laid out at their
 *                         GLOBAL size and this rank fills only its own slots (the rest is zero): the element-wise sum
 *                         of all ranks' top-node bytes is the finished node. NULL layout == pcv_build_octree.
 * The caller's device buffers must stay valid and no other call may be made on the context in between; after a
 * failed pcv_build_finish the tree can only be freed. */
typedef struct pcv_top_streams {
  uint64_t l1[8];          /* |pre(r_c)|: points the level-1 node c holds before its own promotion */
  uint64_t l2[64];         /* |pre(r_cd)| at index c * 8 + d */
  uint32_t l1_split_mask;  /* bit c: level-1 node c is an inner node here */
  uint32_t reserved;
} pcv_top_streams;
typedef struct pcv_top_layout {
  uint64_t root_points;    /* global |pre(root)| == points the root keeps */
  uint64_t l1_stream[8];   /* global |pre(r_c)| */
  uint32_t l1_offset[8];   /* offset of r_c's promoted segment inside pre(root) */
  uint32_t l2_offset[64];  /* offset of r_cd's promoted segment inside pre(r_c) */
} pcv_top_layout;
int pcv_build_begin(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points, pcv_octree** out);
/* pcv_build_begin for points that crossed the exchange as their level-1 chain state (pcv_route_buckets with a
 * pcv_route_state): the root octant digit and the Float32 level-1 codes — with the colour 16 B per point in four 4-byte
 * planes instead of 27 B. The chain continues at level 2 from decode(code) — the position the sending rank held after level 1,
 * so everything downstream is bit-identical. Device memory only; params carry the GLOBAL bounding box. */
typedef struct pcv_routed_points {
  uint64_t n;
  const uint32_t* cx;      /* Float32 bit patterns of the level-1 codes (codec.rs:102-121 with the level-1 child cube) */
  const uint32_t* cy;
  const uint32_t* cz;
  const uint32_t* oct_rgb; /* level-1 digit (0..7) in byte 0, r, g, b in bytes 1..3 */
  const float* intensity;  /* NULL = no "intensity" attribute */
} pcv_routed_points;
int pcv_build_begin_routed(pcv_ctx* ctx, const pcv_build_params* params, const pcv_routed_points* routed, pcv_octree** out);
int pcv_build_top_streams(const pcv_octree* tree, pcv_top_streams* out);
int pcv_build_finish(pcv_octree* tree, const pcv_top_layout* top /* nullable */);

/* One finished node == one `proto::OctreeNode` (proto.proto:90-94) + where its bytes are. */
typedef struct pcv_node_info {
  uint64_t id_high;     /* NodeId u128 halves (src/octree/node.rs:101-111): level<<56 | index>>64 */
  uint64_t id_low;
  int64_t num_points;   /* may be 0 (node exists in meta, no files; generation.rs:241-243) */
  uint32_t level;
  uint32_t encoding;    /* PCV_ENC_* */
  double cube_min[3];   /* NodeId::find_bounding_cube (node.rs:157-172) */
  double cube_edge;
  uint64_t xyz_offset;  /* byte offset of this node's .xyz content inside the xyz blob */
  uint64_t point_offset;/* index of this node's first point inside the rgb / intensity blobs */
} pcv_node_info;

uint64_t pcv_octree_num_nodes(const pcv_octree* t);
uint64_t pcv_octree_num_points(const pcv_octree* t);
int pcv_octree_has_intensity(const pcv_octree* t);
int pcv_octree_node(const pcv_octree* t, uint64_t i, pcv_node_info* out); /* i in (level, index) order */
void pcv_octree_meta(const pcv_octree* t, double* resolution, double bbox_min[3], double bbox_max[3], int* version);
/* File content of node i. which: 0 = .xyz, 1 = .rgb, 2 = .intensity. The host pointer stays valid
 * until pcv_octree_free. Replaces Octree::get_node_data's reads (src/octree/mod.rs:285-307). */
int pcv_octree_node_data(pcv_octree* t, uint64_t i, int which, const uint8_t** data, uint64_t* len);
/* Device-side blobs (no copy): which as above. */
int pcv_octree_device_blob(const pcv_octree* t, int which, const void** dptr, uint64_t* len);
/* Copy the bytes of node i (which: 0 .xyz, 1 .rgb, 2 .intensity) out of the device blob into `dst` (host or device
 * memory, `capacity` bytes) without staging the whole octree on the host. A copy to device memory is only queued on
 * the context's stream (follow with pcv_ctx_synchronize or stream-ordered work); a copy to host memory is complete on
 * return. */
int pcv_octree_copy_node(const pcv_octree* tree, uint64_t i, int which, void* dst, uint64_t capacity, int mem);
/* The same for a list of nodes into ONE destination buffer: copies[k].dst_offset[which] is where node copies[k].node's
 * .xyz / .rgb / .intensity bytes go (UINT64_MAX: skip that file kind). The multi-GPU build uses it to lay its share of
 * the root / level-1 nodes out for the top all-reduce in one call. Same completion rules as pcv_octree_copy_node. */
typedef struct pcv_node_copy {
  uint64_t node;
  uint64_t dst_offset[3];
} pcv_node_copy;
int pcv_octree_copy_nodes(const pcv_octree* tree, const pcv_node_copy* copies, uint64_t count, void* dst, uint64_t capacity, int mem);
/* Write `<NodeId>.xyz/.rgb/.intensity` + meta.pb (version 13) exactly as the reference lays them out
 * (src/read_write/raw.rs:374-449, node_writer.rs:78-89, generation.rs:390-402). */
int pcv_octree_write_dir(pcv_octree* t, const char* directory);
/* Multi-GPU output: every rank writes the node files of its own subtrees (level >= min_level, no meta.pb) ... */
int pcv_octree_write_nodes(pcv_octree* t, const char* directory, uint32_t min_level);
/* ... and one rank writes meta.pb for the gathered node table (id_high, id_low, num_points, encoding are used).
 * Host only; returns PCV_E_IO when the file cannot be written. Layout: proto.proto:58-149, octree/mod.rs:87-99. */
int pcv_write_meta(const char* directory, double resolution, const double bbox_min[3], const double bbox_max[3],
                   const pcv_node_info* nodes, uint64_t count);
void pcv_octree_free(pcv_octree* t);

/* Milliseconds spent per stage of the last pcv_build_octree on this tree (HIP events on the ctx
 * stream). Index with PCV_STAGE_*; returns the number of stages filled. */
#define PCV_STAGE_AABB 0
#define PCV_STAGE_CHAIN_KEYS 1
#define PCV_STAGE_SORT_KEYS 2
#define PCV_STAGE_NODE_SPLIT 3
#define PCV_STAGE_TABLE 4      /* node-table D2H + host finalize + H2D */
#define PCV_STAGE_LEAF_ENCODE 5
#define PCV_STAGE_SORT_RECORDS 6 /* the record sort's FIRST pass (+ histograms); a sort that runs to its end on its own: all passes */
#define PCV_STAGE_PROMOTE_ENCODE 7 /* from the node tables to the finished blobs; CONTAINS PCV_STAGE_SORT_SECOND */
/* The record sort's second pass when it is held back until the node tables are up (it then settles the leaves' points itself,
 * pcv_octree_settled_in_sort): queued inside PCV_STAGE_PROMOTE_ENCODE, measured on its own here — sort time =
 * SORT_RECORDS + SORT_SECOND, promotion proper = PROMOTE_ENCODE - SORT_SECOND. 0 when no pass was held back. */
#define PCV_STAGE_SORT_SECOND 8
#define PCV_STAGE_TOTAL 9
#define PCV_NUM_STAGES 10
int pcv_octree_stage_ms(const pcv_octree* t, float* ms, int cap);
/* How the last build went: attempts == 0: single-chain build (key_levels = levels the sample keys covered);
 * attempts == 1: exact pipeline, the sampled depth estimate held (key_levels = digit levels sorted);
 * attempts >= 2: something was redone (single-chain prediction too shallow and/or depth estimate too shallow). */
void pcv_octree_build_info(const pcv_octree* t, int* key_levels, int* attempts);
/* Single-chain build statistics of the last build (zeros otherwise): nodes and leaves of the predicted tree, points
 * whose leaf is an unsplit candidate node (the codes they kept there are their leaf codes), points that replayed the
 * chain from their coordinates. pcv_octree_spec_continued: points whose chain was continued from the codes kept at a
 * candidate node that turned out to be split. */
void pcv_octree_spec_stats(const pcv_octree* t, uint64_t stats[4]);
uint64_t pcv_octree_spec_continued(const pcv_octree* t);
/* Single-chain build with 12-byte records: number of points whose record left the chain pass with the codes of a
 * Float32-coded level (they wait in a dense side pool, the record names the entry); 0 otherwise. */
uint64_t pcv_octree_wide_pool_entries(const pcv_octree* t);
/* Single-chain build: points of the leaves whose final bytes the record sort's second pass produced itself (integer-coded
 * leaves whose records hold their leaf codes; generation.rs:222-238 — the reference rewrites every point that stays in a
 * node once); the other leaves' records are finished by the settle kernel. 0 when the sort ran to its end on its own. */
uint64_t pcv_octree_settled_in_sort(const pcv_octree* t);
/* Bytes of one record of the last build's record sort (rank + leaf codes + colour): 20, or 12 when the single-chain
 * build packed the record (16-bit codes; the points of Float32-coded levels travel as the index of their pool entry). */
int pcv_octree_record_bytes(const pcv_octree* t);

/* ---- stage-level entry points (unit parity against the oracle) ------------------------------ */
/* K1: find_bounding_box (generation.rs:256-270; Aabb::grow aabb.rs:41-44). n == 0 -> Aabb::zero(). */
int pcv_aabb_reduce(pcv_ctx* ctx, const pcv_points* points, double bbox_min[3], double bbox_max[3]);

/* Per-level table: edge[k] = root_edge / 2^k (node.rs:161), encoding[k] = PositionEncoding::new
 * (src/read_write/codec.rs:31-40). Returns max_level = first k >= 1 with edge[k] <= resolution
 * (no node below it can be split, generation.rs:137), capped at `cap`. */
int pcv_level_table(const double bbox_min[3], const double bbox_max[3], double resolution, int cap, double* edge,
                    int32_t* encoding);
/* The per-level shortcuts the single chain pass takes for this cube (host tables, for the tests that replay them in
 * exact arithmetic): digit_mode[k] = how the octant digit of level k + 1 (node.rs:34-42) is read off the level-k codes
 * (0 = comparison against the centre, 1 = from integer codes, 2 = from Float32 codes); code_threshold[k] = the power of
 * two from which on a Float32 code of level k + 1 (codec.rs:115-121) is taken as 2 v - bit from the level-k code v
 * instead of through the divide / cast chain (0.0 = the step always runs in full). Arrays of PCV_MAX_KEY_LEVELS + 2
 * entries. Returns max_level as pcv_level_table. */
int pcv_level_shortcuts(const double bbox_min[3], const double bbox_max[3], double resolution, uint32_t* digit_mode,
                        double* code_threshold);

/* K2: per-point path digits through the quantise->decode chain (ChildIndex::from_bounding_cube
 * node.rs:34-42, encode codec.rs:102-121, decode codec.rs:124-139, cube recurrence node.rs:157-172).
 * keys[i] has the digit of level k in bits [3*(21-k), 3*(21-k)+3), k = 1..nlevels. */
int pcv_chain_keys(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points, int nlevels,
                   uint64_t* keys /* same memory space as points */);

/* Multi-GPU routing (skew remedy): bucket[i] = 8 * d1 + d2, the level-1 and level-2 octant digits of
 * point i along the same quantise->decode chain K2 follows (ChildIndex::from_bounding_cube node.rs:34-42 against the
 * GLOBAL root cube, then against the level-1 cube after one encode/decode step); counts[b] = points in bucket b.
 * The ranks all-reduce the 64 counts, decide which level-1 nodes the global tree splits, and bin-pack the buckets
 * (whole octants where the level-1 node stays a leaf) onto ranks. Device-resident points only; bucket is a device
 * buffer of n u32, counts a host array of 64 entries. */
typedef struct pcv_route_state {  /* optional outputs of pcv_route_buckets: the level-1 chain state of every point */
  uint32_t* cx;                  /* device, n x u32: Float32 bit patterns of the level-1 codes */
  uint32_t* cy;
  uint32_t* cz;
  uint32_t* oct_rgb;             /* device, n x u32: level-1 digit | r << 8 | g << 16 | b << 24 (needs points->color) */
} pcv_route_state;
int pcv_route_buckets(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points, uint32_t* bucket,
                      uint64_t counts[64], const pcv_route_state* state /* nullable; needs a Float32-encoded level 1 */);

/* The same routing in two passes that never write the level-1 state in input order (69 instead of 87 bytes of traffic per
 * point; a Float32-encoded level 1 is required, as for pcv_route_state):
 *   pcv_route_plan     bucket BYTE of every point (device, n bytes), the bucket histogram of every tile of 4 096 points
 *                      (device, pcv_route_tiles(n) x 64 x u16) and the 64 counts the ranks all-gather;
 *   pcv_route_scatter  once the plan (bucket -> owning rank) is known: the level-1 state of every point — the four planes of
 *                      pcv_route_state, plus points->intensity when set — computed from the coordinates again and written
 *                      straight to the point's place in its owner's buffer: row k (in input order) of the rows owned by rank
 *                      r goes to row k of dst[r]'s planes. Replaces ChildIndex::from_bounding_cube (node.rs:34-42) + the
 *                      first encode step (codec.rs:102-121) of generation.rs:78-99 for the multi-GPU exchange. */
/* pcv_route_plan, params->flags: ownership goes by ROOT OCTANT (BASELINE north_star: shard by the top-3-bit prefix) — the
 * bucket of a point is its level-1 digit alone (bucket = d1 << 3, the counts of the other buckets are 0): three comparisons
 * against the root cube's centre per point (node.rs:34-42) instead of a level step of the chain. pcv_route_scatter is the same. */
#define PCV_ROUTE_OCTANTS_ONLY 64u
typedef struct pcv_route_dst {
  uint32_t* oct_rgb;
  uint32_t* cx;
  uint32_t* cy;
  uint32_t* cz;
  float* intensity; /* nullable */
} pcv_route_dst;
uint64_t pcv_route_tiles(uint64_t n);
int pcv_route_plan(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points, uint8_t* bucket, uint16_t* tile_hist,
                   uint64_t counts[64]);
int pcv_route_scatter(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points, const uint8_t* bucket,
                      const uint16_t* tile_hist, uint32_t world, const uint8_t rank_of_bucket[64], const pcv_route_dst* dst /* [world] */);

/* Stable partition of up to 8 row-aligned planes by owner (owner[i] is a rank, or a bucket when rank_of_bucket maps the
 * 64 buckets to ranks): row k (in input order) of the rows owned by rank r goes to row k of dst[r * nplanes + p] for
 * every plane p. The caller points dst[r * nplanes + p] at its send buffer for rank r, and the own rank's entries
 * straight at the receive buffer. All pointers are device pointers; rows are 1..16 bytes. */
typedef struct pcv_plane {
  const void* src;
  uint32_t elem_bytes;
} pcv_plane;
int pcv_partition_by_owner(pcv_ctx* ctx, uint64_t n, const uint32_t* owner, uint32_t world,
                           const uint8_t* rank_of_bucket /* nullable host array of 64 */, uint32_t nplanes,
                           const pcv_plane* planes, void* const* dst /* [world][nplanes] */);

/* K4: the octree topology from SORTED path keys (split / should_split_node / split_node, generation.rs:58-193): a child
 * exists iff a key carries its prefix, a child is split iff count > max_points_per_node && child edge > resolution,
 * the root is always split. keys: pcv_chain_keys layout, ascending, full depth of params' level table (<= 21 levels).
 * nodes come back breadth first (level-major, prefix order inside a level), the children of a node consecutive in digit
 * order. *num_nodes may exceed `capacity`: only the first `capacity` are written. PCV_E_DEPTH: a node at the last key
 * level would still have to be split. */
typedef struct pcv_split_node {
  uint64_t id_high, id_low; /* NodeId halves (node.rs:101-111) */
  uint64_t first, count;    /* the subtree's points: a contiguous range of the key-sorted order */
  uint32_t level;
  uint32_t parent;          /* index in this table, 0xffffffff for the root */
  uint32_t first_child;     /* index of the first child (inner nodes) */
  uint32_t child_mask;      /* bit c: child c exists */
  uint32_t is_leaf;
  uint32_t reserved;
} pcv_split_node;
int pcv_node_split(pcv_ctx* ctx, const pcv_build_params* params, const uint64_t* sorted_keys, uint64_t n, int mem,
                   pcv_split_node* nodes, uint64_t capacity, uint64_t* num_nodes);

/* K5 (table part): the closed form of subsample_children_into (generation.rs:195-253, 335-387) on a node
 * table: stream_len = |pre(node)| (leaves: their points; inner: sum over children of ceil(|pre(child)| / 8)),
 * num_points = what the node keeps (root: everything it receives; others: |pre| - ceil(|pre| / 8)), child_offset =
 * where the node's promoted block starts inside its parent's stream. With node_of_slot / slot_in_node (nullable, n
 * entries): the final home of the record at every position of the leaf-sorted order — it climbs while its position j in
 * the current stream is a multiple of 8 (j' = child_offset + j / 8), otherwise it settles at slot j - j / 8 - 1
 * (the root keeps slot j). Host only. */
typedef struct pcv_promote_node {
  uint64_t stream_len;
  uint64_t num_points;
  uint64_t child_offset;
} pcv_promote_node;
int pcv_promote_assign(const pcv_split_node* nodes, uint64_t num_nodes, pcv_promote_node* per_node, uint64_t n,
                       uint32_t* node_of_slot, uint32_t* slot_in_node);

/* K5 + K6: leaf encode, stable grouping by leaf, promotion and final encode for a GIVEN topology (a node table as
 * pcv_node_split returns it, built for the same points / params): the finished octree, as pcv_build_octree returns it.
 * With pcv_chain_keys + pcv_sort_keys64 + pcv_node_split this is the whole build, stage by stage. */
int pcv_gather_encode(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points, const pcv_split_node* nodes,
                      uint64_t num_nodes, pcv_octree** out);

/* K3: stable LSD radix sort of 64-bit keys on bits [begin_bit, end_bit), in place. */
int pcv_sort_keys64(pcv_ctx* ctx, uint64_t* keys, uint64_t n, int begin_bit, int end_bit, int mem);
/* K3: the 32-bit key variant the build uses when ten levels of path digits suffice. */
int pcv_sort_keys32(pcv_ctx* ctx, uint32_t* keys, uint64_t n, int begin_bit, int end_bit, int mem);
/* K3: stable sort of (u32 key, u32 value) pairs on bits [begin_bit, end_bit), in place. */
int pcv_sort_pairs32(pcv_ctx* ctx, uint32_t* keys, uint32_t* values, uint64_t n, int begin_bit, int end_bit,
                     int mem);

/* Device self-test of the exact constant-divisor division used by the chain kernels (see pcv_chain_dev.h):
 * compares it bit-for-bit with IEEE f64 division for every integer code / 255 and / 65535 and for
 * samples_per_divisor pseudo-random numerators per divisor; *mismatches must come back 0. */
int pcv_selftest_division(pcv_ctx* ctx, const double* divisors, int ndiv, uint64_t samples_per_divisor,
                          uint64_t* mismatches);

/* ---- PLY ingest (host side) --------------------------------------------------------- */
/* Replaces PlyIterator (src/read_write/ply.rs:328-556): binary little-endian PLY, element `vertex` first; x/y/z of
 * any scalar type cast to f64 plus the header's `comment offset: x y z`; red/green/blue (uchar) -> colour;
 * `intensity` (float) kept; alpha and all other properties skipped. One pass; the arrays can go straight into
 * pcv_build_octree with PCV_BUILD_COMPUTE_BBOX (== build_octree_from_file, generation.rs:272-287).
 * `err` (nullable) receives a message on failure. */
typedef struct pcv_ply pcv_ply;
int pcv_ply_read(const char* path, pcv_ply** out, char* err, uint64_t errcap);
uint64_t pcv_ply_num_points(const pcv_ply* ply);
/* Fills `out` with host pointers owned by `ply` (color / intensity are NULL when the file has none). */
int pcv_ply_points(const pcv_ply* ply, pcv_points* out);
void pcv_ply_free(pcv_ply* ply);
/* build_octree_from_file (src/octree/generation.rs:272-287) with the decode on the device: the vertex records of the
 * file go up as they are (15 bytes per point for float x y z + uchar r g b instead of the 27 of f64 SoA arrays), one HIP
 * kernel casts x / y / z to f64 and adds the header offset like ply.rs:488-493, then the build runs with
 * PCV_BUILD_COMPUTE_BBOX (params->bbox_* are ignored). Same files as pcv_ply_read + pcv_build_octree; the PLY must
 * have colour, and `intensity` (float) when with_intensity != 0. */
int pcv_build_octree_from_ply(pcv_ctx* ctx, const pcv_build_params* params, const char* path, int with_intensity,
                              pcv_octree** out);

/* ---- octree loading (viewer side) ------------------------------------------------------------ */
/* Replaces Octree::from_data_provider over an OnDiskDataProvider (src/octree/mod.rs:156-215,
 * src/data_provider/on_disk.rs): parses meta.pb — versions 9..13 like the reference (9-11: top-level
 * deprecated_resolution / deprecated_nodes, Vector3f boxes and level/index NodeIds where present; 12: the box inside
 * OctreeMeta; 13: current), anything else is InvalidVersion — and derives every node's bounding cube
 * (NodeId::find_bounding_cube). Node files are read on demand by pcv_octree_node_data; the first point query
 * (pcv_query_points / pcv_cull_node_points) reads all node files once and keeps them device resident. */
int pcv_octree_open_dir(pcv_ctx* ctx, const char* directory, pcv_octree** out);

/* ---- queries: batched transform-and-cull ------------------------------------------------------ */
/* PointLocation variants (src/iterator.rs:12-20) that the octree path supports. params layout:
 *   PCV_SHAPE_ALL                   -                                          AllPoints
 *   PCV_SHAPE_AABB                  min xyz, max xyz                           geometry::Aabb
 *   PCV_SHAPE_FRUSTUM               clip_from_query (16, nalgebra memory order = column-major); the inverse is
 *                                   computed like Frustum::from_matrix4 (src/geometry/frustum.rs:111-117)
 *   PCV_SHAPE_FRUSTUM_WITH_INVERSE  clip_from_query (16) then query_from_clip (16), as Frustum::new stores them
 *                                   (frustum.rs:101-108)
 *   PCV_SHAPE_OBB                   query_from_obb isometry: translation xyz, unit quaternion i j k w; then the
 *                                   half extent xyz (src/geometry/obb.rs:13-45) */
#define PCV_SHAPE_ALL 0
#define PCV_SHAPE_AABB 1
#define PCV_SHAPE_FRUSTUM 2
#define PCV_SHAPE_OBB 3
#define PCV_SHAPE_FRUSTUM_WITH_INVERSE 4
typedef struct pcv_shape {
  int32_t kind;
  int32_t reserved;
  double params[32];
} pcv_shape;
typedef struct pcv_shapes pcv_shapes;

/* Relation (src/math/sat.rs:37-45) */
#define PCV_REL_IN 0
#define PCV_REL_CROSS 1
#define PCV_REL_OUT 2

/* Q1: prepares `count` shapes on the device: corners, unique edges / face normals and the deduplicated
 * separating axes against AABBs (Intersector::cache_separating_axes_for_aabb, src/math/sat.rs:111-143). */
int pcv_shapes_create(pcv_ctx* ctx, const pcv_shape* shapes, uint32_t count, pcv_shapes** out);
void pcv_shapes_free(pcv_shapes* shapes);
uint32_t pcv_shapes_count(const pcv_shapes* shapes);
/* Inspect one prepared shape (tests): 8 corners, up to 26 axes. valid == 0: the matrix is not invertible. */
int pcv_shapes_get(pcv_shapes* shapes, uint32_t i, double corners[24], double axes[78], uint32_t* num_axes, int* valid);

/* Q2: Relation of every node cube against every shape (CachedAxesIntersector::intersect, sat.rs:167-194), row
 * major [shape][node] (node order as pcv_octree_node), host buffers. size_on_screen (nullable, same shape) is
 * relative_size_on_screen (src/octree/mod.rs:119-139) for the shape's clip_from_query; NaN where w == 0. */
int pcv_cull_nodes(pcv_ctx* ctx, const pcv_shapes* shapes, pcv_octree* tree, uint8_t* relation, double* size_on_screen);
/* Q2 as a list per shape (round 5): the nodes whose Relation is not Out (sat.rs:174-194), in node order —
 * node_indices / relation / size_on_screen are [shape][capacity] host arrays, counts[shape] the number of such nodes
 * (entries past `capacity` are dropped, the count is not). size_on_screen (nullable) is relative_size_on_screen
 * (octree/mod.rs:119-139), computed for the listed nodes only — the nodes the reference projects (octree/mod.rs:261-272).
 * The same Relations as pcv_cull_nodes without its shapes x nodes matrix (config 4: 60.7 M pairs, 0.24 % not Out).
 * Round 6: found by a walk down the tree, like the reference's own traversals (octree_iterator.rs:30-43) — a subtree is
 * skipped only under a node that is Out by a margin far above the rounding error of its descendants' bounds, and any shape
 * that meets an Out node without that margin is evaluated flat: the lists equal pcv_cull_nodes' rows in every case. */
int pcv_cull_nodes_sparse(pcv_ctx* ctx, const pcv_shapes* shapes, pcv_octree* tree, uint32_t capacity, uint32_t* counts,
                          uint32_t* node_indices, uint8_t* relation, double* size_on_screen);

/* Q3: Octree::get_visible_nodes (src/octree/mod.rs:228-283) for every frustum: node indices in the order the
 * reference's BinaryHeap pops them. counts[f] = number of visible nodes (may exceed `capacity`; only the first
 * `capacity` are written to node_indices[f * capacity ..]). status[f]: 0 ok, 1 matrix not invertible
 * (the reference panics), 2 a projected corner had w == 0 (the reference panics). */
int pcv_visible_nodes(pcv_ctx* ctx, const pcv_shapes* frusta, pcv_octree* tree, uint32_t capacity, uint32_t* counts,
                      uint32_t* node_indices, int32_t* status);
/* PointCloud::nodes_in_location (src/octree/mod.rs:309-331, src/octree/octree_iterator.rs): breadth-first, a node
 * is reported and descended into iff its cube is not Relation::Out. */
int pcv_nodes_in_location(pcv_ctx* ctx, const pcv_shapes* shapes, pcv_octree* tree, uint32_t capacity, uint32_t* counts,
                          uint32_t* node_indices);

/* Q4: FilteredIterator's keep mask (src/iterator.rs:96-119): shape.contains(p) AND, when `interval` is given,
 * interval[0] <= intensity <= interval[1] (ClosedInterval, src/math/mod.rs:86-88). keep lives where the points live;
 * kept (nullable) receives the number of ones. */
int pcv_cull_points(pcv_ctx* ctx, const pcv_shapes* shapes, uint32_t shape_index, const pcv_points* points,
                    const double* interval, uint8_t* keep, uint64_t* kept);
/* Same on a built octree's node: positions are decoded on the fly from the node's device-resident bytes
 * (src/read_write/codec.rs:124-139); keep is a host buffer of num_points bytes. */
int pcv_cull_node_points(pcv_ctx* ctx, const pcv_shapes* shapes, uint32_t shape_index, pcv_octree* tree, uint64_t node,
                         const double* interval, uint8_t* keep, uint64_t* kept);

/* Batched point query — the per-location work of ParallelIterator::try_for_each_batch
 * (src/iterator.rs:255-333): nodes_in_location, then for every reported node the FilteredIterator keep mask on the
 * node's decoded positions, then `retain` — here as a stable compaction in (node traversal order, point order).
 * Outputs (capacity entries each; `mem` says where they live): decoded f64 x/y/z, rgb (3 B per point) and, when
 * non-null and the octree has it, intensity. *count = number of points that passed (may exceed capacity: only the
 * first `capacity` are written). Works on built octrees and on octrees opened with pcv_octree_open_dir (the
 * reference's use: stream_points_for_query_in_node -> points_in_node -> NodeIterator over node files,
 * src/iterator.rs:185-223, src/octree/mod.rs:285-307, src/read_write/node_iterator.rs:24-119). */
int pcv_query_points(pcv_ctx* ctx, const pcv_shapes* shapes, uint32_t shape_index, pcv_octree* tree, const double* interval,
                     uint64_t capacity, int mem, double* x, double* y, double* z, uint8_t* rgb, float* intensity,
                     uint64_t* count);

/* The same for ONE node: PointCloud::stream_points_for_query_in_node (src/iterator.rs:185-205) — the points of node
 * `node` (index as pcv_octree_node) that pass the shape and the interval, in file order. ParallelIterator hands the nodes
 * of nodes_in_location to its workers one by one; a veneer that keeps that structure calls this per node. */
int pcv_query_node_points(pcv_ctx* ctx, const pcv_shapes* shapes, uint32_t shape_index, pcv_octree* tree, uint64_t node,
                          const double* interval, uint64_t capacity, int mem, double* x, double* y, double* z, uint8_t* rgb,
                          float* intensity, uint64_t* count);

/* The `/nodes_data` reply blob of octree_web_viewer (octree_web_viewer/src/backend.rs:90-177) for a list of nodes:
 * per node min xyz (3 x f64 LE), edge (f64), num_points (u32), bytes per coordinate (u8), pad to 8, raw .xyz, pad
 * to 8, raw .rgb, pad to 8. *needed = blob size; the blob is written when out != NULL and capacity >= *needed. */
int pcv_octree_nodes_blob(pcv_octree* t, const uint64_t* node_indices, uint64_t count, uint8_t* out, uint64_t capacity,
                          uint64_t* needed);

/* Q5: Isometry3 * Point3 for a batch (xray/src/generation.rs:493-497; Aabb::transform aabb.rs:58-66 uses the
 * same product). iso = translation xyz, unit quaternion i j k w. Outputs live where the inputs live. */
int pcv_transform_points(pcv_ctx* ctx, const double iso[7], const pcv_points* points, double* ox, double* oy, double* oz);

#ifdef __cplusplus
}
#endif
#endif /* PCV_HIP_H */
