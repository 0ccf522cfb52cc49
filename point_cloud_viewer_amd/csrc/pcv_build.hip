// pcv_build.hip — host orchestration of the device-resident octree build behind the C ABI (include/pcv_hip.h).
//
// Replaces build_octree (reference src/octree/generation.rs:289-403). The reference streams every point
// through one file per node per level; here the tree is derived from a stable sort of chain-exact path keys:
//
//   K1 aabb_reduce (optional)            find_bounding_box                   generation.rs:256-270
//   K2 chain_keys                        per-level octant digits             generation.rs:78-83 + codec.rs
//   K3 sort keys                         (groups points by path)             generation.rs:84-101
//   K4 node_split                        should_split_node / recursion       generation.rs:110-193
//   host: node table finalize            counts of the promotion pyramid     generation.rs:195-253 (closed form)
//   K5 leaf_encode                       leaf-level quantisation             node_writer.rs:281-316
//   K3 sort records by leaf rank         stable => per-node input order      SURVEY F11
//   K6 promote_encode                    every-8th promotion + rewrites      generation.rs:222-238
//
// Everything between the first and last kernel stays in HBM; the only host round trip is the (small) node
// table. No CPU fallback exists: if HIP fails the call fails.
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstring>

#include "pcv_internal.h"
#include "pcv_spec.h"

// ------------------------------------------------------------------------------------------------
// context, pool
// ------------------------------------------------------------------------------------------------
void* PcvPool::alloc(size_t bytes, hipError_t* err) {
  *err = hipSuccess;
  if (bytes == 0) bytes = 256;
  bytes = (bytes + 255) & ~(size_t)255;
  // Reuse a cached block only if it is about the requested size: a loose match lets a small request grab a big block
  // and the big request that follows pays a multi-millisecond hipMalloc in the middle of a build.
  auto it = free_blocks.lower_bound(bytes);
  if (it != free_blocks.end() && it->first <= bytes + bytes / 8 + (64u << 10)) {
    void* p = it->second;
    live[p] = it->first;
    free_blocks.erase(it);
    return p;
  }
  void* p = nullptr;
#ifdef PCV_EXPERIMENTS  // placement experiments (profiles/r05_placement_probe.json): neither helps, neither ships
  // PCV_POOL_VMM=<chunk MiB> (libpcv_hip_exp.so, tools/placement_probe.py): big blocks assembled from physical chunks of that
  // size mapped in a scrambled order. The scatter kernels of the build run 1.0-1.25 ms on whatever hipMalloc returns and
  // 1.36-1.53 ms on physically CONTIGUOUS memory (PCV_POOL_CONTIG=1): their 131 072 write streams meet in the same
  // memory channels when the physical addresses follow the virtual ones too regularly.
  static const size_t vmm_chunk = [] {
    const char* e = pcv_experiment("PCV_POOL_VMM");
    return e ? (size_t)std::max(0, atoi(e)) << 20 : (size_t)0;
  }();
  if (vmm_chunk && bytes >= (64u << 20)) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran) {
      const size_t chunk = (vmm_chunk + gran - 1) / gran * gran;
      const size_t size = (bytes + chunk - 1) / chunk * chunk;
      const size_t nchunks = size / chunk;
      void* va = nullptr;
      PcvVmmBlock blk;
      blk.size = size;
      bool ok = hipMemAddressReserve(&va, size, 0, nullptr, 0) == hipSuccess;
      for (size_t k = 0; ok && k < nchunks; ++k) {
        hipMemGenericAllocationHandle_t h;
        ok = hipMemCreate(&h, chunk, &prop, 0) == hipSuccess;
        if (ok) blk.handles.push_back(h);
      }
      if (ok) {
        // slot k of the address range gets chunk perm(k): a multiplicative scramble (odd multiplier modulo a power of two,
        // values past nchunks skipped) — deterministic, no two slots share a chunk
        size_t pow2 = 1;
        while (pow2 < nchunks) pow2 <<= 1;
        size_t slot = 0;
        for (size_t v = 0; ok && v < pow2; ++v) {
          const size_t c = (v * 0x9E3779B1ull + 12345u) & (pow2 - 1);
          if (c >= nchunks) continue;
          ok = hipMemMap((char*)va + slot * chunk, chunk, 0, blk.handles[c], 0) == hipSuccess;
          ++slot;
        }
        ok = ok && slot == nchunks;
      }
      if (ok) {
        hipMemAccessDesc ad = {};
        ad.location.type = hipMemLocationTypeDevice;
        ad.location.id = device;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        ok = hipMemSetAccess(va, size, &ad, 1) == hipSuccess;
      }
      if (ok) {
        vmm[va] = blk;
        live[va] = bytes;
        return va;
      }
      (void)hipGetLastError();
      if (va) {
        (void)hipMemUnmap(va, size);
        for (auto h : blk.handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(va, size);
      }
    }
  }
  // PCV_POOL_CONTIG=1 (libpcv_hip_exp.so, tools/placement_probe.py): big blocks asked for as physically contiguous memory
  static const bool contig = [] {
    const char* e = pcv_experiment("PCV_POOL_CONTIG");
    return e && atoi(e) != 0;
  }();
  if (contig && bytes >= (8u << 20)) {
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) == hipSuccess) {
      live[p] = bytes;
      return p;
    }
    (void)hipGetLastError();
    p = nullptr;
  }
#endif
  *err = hipMalloc(&p, bytes);
  if (*err != hipSuccess) {
    // drop the cache and retry once
    trim();
    *err = hipMalloc(&p, bytes);
    if (*err != hipSuccess) return nullptr;
  }
  live[p] = bytes;
  return p;
}
void PcvPool::release(void* p) {
  if (!p) return;
  auto it = live.find(p);
  if (it == live.end()) return;
  free_blocks.insert({it->second, p});
  live.erase(it);
}
void PcvPool::free_block(void* p) {
  auto it = vmm.find(p);
  if (it == vmm.end()) {
    (void)hipFree(p);
    return;
  }
  (void)hipMemUnmap(p, it->second.size);
  for (auto h : it->second.handles) (void)hipMemRelease(h);
  (void)hipMemAddressFree(p, it->second.size);
  vmm.erase(it);
}
void PcvPool::trim() {
  for (auto& kv : free_blocks) free_block(kv.second);
  free_blocks.clear();
}

int pcv_ctx::dev_alloc(void** p, size_t bytes) {
  hipError_t e;
  *p = pool.alloc(bytes, &e);
  if (!*p) return fail(e == hipErrorOutOfMemory ? PCV_E_OOM : PCV_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  return PCV_OK;
}
void pcv_ctx::dev_free(void* p) { pool.release(p); }
int pcv_ctx::host_alloc(void** p, size_t bytes) {
  if (bytes == 0) bytes = 256;
  auto it = host_free.lower_bound(bytes);
  if (it != host_free.end() && it->first <= bytes * 2 + (1u << 20)) {
    *p = it->second;
    host_live[*p] = it->first;
    host_free.erase(it);
    return PCV_OK;
  }
  hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
  if (e != hipSuccess) return fail(PCV_E_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  host_live[*p] = bytes;
  return PCV_OK;
}
void pcv_ctx::host_release(void* p) {
  if (!p) return;
  auto it = host_live.find(p);
  if (it == host_live.end()) return;
  host_free.insert({it->second, p});
  host_live.erase(it);
}
int pcv_ctx::pinned_reserve(size_t bytes) {
  if (bytes <= pinned_bytes) return PCV_OK;
  if (pinned) (void)hipHostFree(pinned);
  pinned = nullptr;
  pinned_bytes = 0;
  hipError_t e = hipHostMalloc(&pinned, bytes, hipHostMallocDefault);
  if (e != hipSuccess) return fail(PCV_E_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  pinned_bytes = bytes;
  return PCV_OK;
}

int pcv_ctx::table_dev_reserve(size_t bytes) {
  if (bytes <= table_dev_bytes) return PCV_OK;
  // the old block may still be read by work in flight: drain both streams before it goes away
  if (table_dev) {
    (void)hipStreamSynchronize(stream);
    (void)hipStreamSynchronize(side);
    (void)hipFree(table_dev);
    table_dev = nullptr;
    table_dev_bytes = 0;
  }
  const size_t want = (bytes + (bytes >> 1) + 4095) & ~(size_t)4095;
  if (hipMalloc(&table_dev, want) != hipSuccess) return fail(PCV_E_OOM, "out of device memory (node tables)");
  table_dev_bytes = want;
  return PCV_OK;
}

int pcv_ctx::pinned_spec_reserve(size_t bytes) {
  if (bytes <= pinned_spec_bytes) return PCV_OK;
  bytes += bytes / 2;  // the size follows the node count of the input: leave room so that similar builds do not regrow it
  if (pinned_spec) (void)hipHostFree(pinned_spec);  // waits for copies in flight
  pinned_spec = nullptr;
  pinned_spec_bytes = 0;
  hipError_t e = hipHostMalloc(&pinned_spec, bytes, hipHostMallocDefault);
  if (e != hipSuccess) return fail(PCV_E_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  pinned_spec_bytes = bytes;
  return PCV_OK;
}

void PcvHostPool::start(unsigned n) {
  if (!threads.empty()) return;
  for (unsigned k = 0; k < n; ++k)
    threads.emplace_back([this] {
      uint64_t seen = 0;
      for (;;) {
        std::unique_lock<std::mutex> lk(mu);
        wake.wait(lk, [&] { return stop || (generation != seen && next < count); });
        if (stop) return;
        while (next < count) {
          const size_t i = next++;
          lk.unlock();
          job(i);
          lk.lock();
          if (++finished == count) done.notify_all();
        }
        seen = generation;
      }
    });
}
void PcvHostPool::run(size_t n, const std::function<void(size_t)>& fn) {
  if (n == 0) return;
  if (threads.empty() || n == 1) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::unique_lock<std::mutex> lk(mu);
  job = fn;
  next = 0;
  count = n;
  finished = 0;
  ++generation;
  wake.notify_all();
  while (next < count) {  // the caller works too
    const size_t i = next++;
    lk.unlock();
    fn(i);
    lk.lock();
    ++finished;
  }
  done.wait(lk, [&] { return finished == count; });
  count = 0;
}
PcvHostPool::~PcvHostPool() {
  {
    std::lock_guard<std::mutex> lk(mu);
    stop = true;
  }
  wake.notify_all();
  for (auto& th : threads) th.join();
}

// Pageable caller memory -> device: the runtime's own staging of a pageable hipMemcpy runs on one thread (~45 GB/s
// here); several host threads filling a ring of pinned chunks keep the link busy instead (every chunk is one DMA).
int pcv_ctx::h2d(void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return PCV_OK;
  if (bytes < (4u << 20)) {  // small arrays: not worth the ring
    PCV_HIP_CHECK(this, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
    return PCV_OK;
  }
  const uint8_t* from = (const uint8_t*)src;
  return h2d_fill(dst, bytes, [from](uint8_t* to, size_t off, size_t len) {
    std::memcpy(to, from + off, len);
    return true;
  });
}

// The ring of pinned chunks and the host threads that fill them, created on first use (h2d_fill, pcv_ingest_begin).
int pcv_ctx::ring_ensure() {
  if (ring[0]) return PCV_OK;
  for (int k = 0; k < kRingSlots; ++k) {
    if (hipHostMalloc(&ring[k], kRingChunk, hipHostMallocDefault) != hipSuccess) return fail(PCV_E_OOM, "hipHostMalloc (staging ring)");
    if (hipEventCreateWithFlags(&ring_ev[k], hipEventDisableTiming) != hipSuccess) return fail(PCV_E_HIP, "hipEventCreate");
  }
  unsigned hw = std::thread::hardware_concurrency();
  // copies into pinned memory saturate the link with 7 threads; preads from a file (pcv_build_octree_from_ply) want more
  unsigned workers = hw >= 64 ? 15 : (hw >= 16 ? 7 : (hw > 2 ? hw / 2 - 1 : 0));
  if (const char* e = pcv_experiment("PCV_H2D_THREADS")) workers = (unsigned)std::max(0, atoi(e));
  host_pool.start(workers);
  return PCV_OK;
}

// Host -> device through the ring of pinned chunks: `fill(to, off, len)` produces bytes [off, off + len) of the source
// into pinned memory (a memcpy from pageable memory, a pread from a file) and is called from the context's host threads,
// 2 MiB per call, several calls in parallel; one DMA per 32 MiB chunk follows. false from `fill` -> PCV_E_IO.
int pcv_ctx::h2d_fill(void* dst, size_t bytes, const std::function<bool(uint8_t*, size_t, size_t)>& fill) {
  if (bytes == 0) return PCV_OK;
  if (int rc = ring_ensure()) return rc;
  // one part per worker (the caller works too) and chunk, not less than 256 KiB
  const size_t nworkers = host_pool.threads.size() + 1;
  const size_t kPart = std::max<size_t>(256u << 10, ((kRingChunk + nworkers - 1) / nworkers + 4095) & ~(size_t)4095);
  std::atomic<int> bad{0};
#ifdef PCV_EXPERIMENTS
  static const bool trace = pcv_experiment("PCV_H2D_TRACE") != nullptr;
  double t_wait = 0, t_fill = 0, t_issue = 0;
  const auto t_begin = std::chrono::steady_clock::now();
#define PCV_H2D_T(acc, stmt)                                                                          \
  {                                                                                                   \
    const auto t0_ = std::chrono::steady_clock::now();                                                \
    stmt;                                                                                             \
    acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count(); \
  }
#else
#define PCV_H2D_T(acc, stmt) stmt;
#endif
  for (size_t off = 0; off < bytes; off += kRingChunk) {
    const size_t len = bytes - off < kRingChunk ? bytes - off : kRingChunk;
    const int slot = ring_take();
    if (ring_busy[slot]) PCV_H2D_T(t_wait, PCV_HIP_CHECK(this, hipEventSynchronize(ring_ev[slot])))  // its previous DMA has left the chunk
    uint8_t* chunk = (uint8_t*)ring[slot];
    PCV_H2D_T(t_fill, host_pool.run((len + kPart - 1) / kPart, [&](size_t p) {
      const size_t b = p * kPart, e = b + kPart < len ? b + kPart : len;
      if (!fill(chunk + b, off + b, e - b)) bad.store(1);
    }))
    if (bad.load()) return fail(PCV_E_IO, "reading the source of a host-to-device copy failed");
    PCV_H2D_T(t_issue, PCV_HIP_CHECK(this, hipMemcpyAsync((uint8_t*)dst + off, chunk, len, hipMemcpyHostToDevice, stream));
              PCV_HIP_CHECK(this, hipEventRecord(ring_ev[slot], stream)))
    ring_busy[slot] = true;
  }
#undef PCV_H2D_T
#ifdef PCV_EXPERIMENTS
  if (trace)
    fprintf(stderr, "[pcv h2d] %.1f MB: total %.2f ms (waiting for a ring slot %.2f, filling %.2f, issuing %.2f)\n", bytes / 1e6,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), t_wait, t_fill, t_issue);
#endif
  return PCV_OK;
}

hipEvent_t pcv_ctx::prof_event() {
  if (!prof_free.empty()) {
    hipEvent_t e = prof_free.back();
    prof_free.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
// Call only after the stream has been synchronised.
void pcv_ctx::prof_resolve() {
  for (auto& p : prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      prof_ms[p.id] += ms;
      prof_launches[p.id] += 1;
    }
    prof_free.push_back(p.a);
    prof_free.push_back(p.b);
  }
  prof_pending.clear();
}

static const char* kKernelNames[PCV_K_COUNT] = {
    "aabb_partial_kernel", "chain_keys_kernel",  "upsweep_kernel<u64>",   "scan_kernel",
    "downsweep_kernel<u64>", "split_search_kernel", "split_assign_kernel", "leaf_encode_kernel",
    "upsweep_kernel<u32>", "downsweep_kernel<u32>", "promote_settle_kernel", "downsweep_rec_kernel", "cull_nodes_kernel",
    "visible_nodes_kernel", "nodes_in_location_kernel", "cull_points_kernel", "transform_points_kernel",
    "query_compact_kernel", "route_bucket_kernel", "partition_count_kernel", "partition_scatter_kernel",
    "promote_climb_kernel", "spec_encode_kernel", "rank_hist_kernel", "spec_continue_kernel", "spec_replay_kernel", "upsweep_map_kernel",
    "hist_from_rows_kernel", "cull_nodes_sparse_kernel", "downsweep_settle_kernel", "ingest_batch_kernel"};
static_assert(sizeof(kKernelNames) / sizeof(kKernelNames[0]) == PCV_K_COUNT, "kernel name table out of sync");

extern "C" int pcv_ctx_set_profiling(pcv_ctx* ctx, int enabled) {
  if (!ctx) return PCV_E_INVALID;
  ctx->profiling = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
  return PCV_OK;
}
extern "C" int pcv_ctx_reset_kernel_stats(pcv_ctx* ctx) {
  if (!ctx) return PCV_E_INVALID;
  for (int i = 0; i < PCV_K_COUNT; ++i) {
    ctx->prof_launches[i] = 0;
    ctx->prof_ms[i] = 0;
  }
  return PCV_OK;
}
extern "C" int pcv_ctx_kernel_stats(pcv_ctx* ctx, int kernel_id, const char** name, uint64_t* launches,
                                    double* total_ms) {
  if (!ctx) return PCV_E_INVALID;
  if (kernel_id < 0 || kernel_id >= PCV_K_COUNT) return PCV_K_COUNT;
  if (!ctx->prof_pending.empty()) {  // launches of the stage-level entry points are resolved on first read
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->prof_resolve();
  }
  if (name) *name = kKernelNames[kernel_id];
  if (launches) *launches = ctx->prof_launches[kernel_id];
  if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
  return PCV_K_COUNT;
}

extern "C" int pcv_abi_version(void) { return PCV_ABI_VERSION; }

extern "C" int pcv_ctx_create(int device, void* stream, pcv_ctx** out) {
  if (!out) return PCV_E_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return PCV_E_HIP;
  if (hipSetDevice(device) != hipSuccess) return PCV_E_HIP;
  pcv_ctx* c = new pcv_ctx();
  c->device = device;
  c->pool.device = device;
  if (hipHostMalloc((void**)&c->mailbox, 136 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) {
    delete c;
    return PCV_E_OOM;
  }
  if (hipHostGetDevicePointer((void**)&c->mailbox_dev, c->mailbox, 0) != hipSuccess) c->mailbox_dev = c->mailbox;
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return PCV_E_HIP;
    }
    c->own_stream = true;
  }
  for (auto& e : c->ev)
    if (hipEventCreate(&e) != hipSuccess) {
      delete c;
      return PCV_E_HIP;
    }
  for (int k = 0; k < PCV_NUM_STAGES; ++k)
    if (hipEventCreate(&c->stage_b[k]) != hipSuccess || hipEventCreate(&c->stage_e[k]) != hipSuccess) {
      delete c;
      return PCV_E_HIP;
    }
  if (hipEventCreateWithFlags(&c->spec_ev, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return PCV_E_HIP;
  }
  if (hipEventCreateWithFlags(&c->xev, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return PCV_E_HIP;
  }
  if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->side_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->side_join, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return PCV_E_HIP;
  }
  *out = c;
  return PCV_OK;
}

extern "C" void pcv_ctx_destroy(pcv_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->pool.trim();
  {
    std::vector<void*> still;
    for (auto& kv : ctx->pool.live) still.push_back(kv.first);
    for (void* q : still) ctx->pool.free_block(q);
  }
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->pinned_spec) (void)hipHostFree(ctx->pinned_spec);
  for (int k = 0; k < pcv_ctx::kRingSlots; ++k) {
    if (ctx->ring[k]) (void)hipHostFree(ctx->ring[k]);
    if (ctx->ring_ev[k]) (void)hipEventDestroy(ctx->ring_ev[k]);
  }
  if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
  for (auto& kv : ctx->host_free) (void)hipHostFree(kv.second);
  for (auto& kv : ctx->host_live) (void)hipHostFree(kv.first);
  for (auto& e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->xev) (void)hipEventDestroy(ctx->xev);
  if (ctx->spec_ev) (void)hipEventDestroy(ctx->spec_ev);
  if (ctx->side) {
    (void)hipStreamSynchronize(ctx->side);
    (void)hipStreamDestroy(ctx->side);
  }
  if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
  if (ctx->side_join) (void)hipEventDestroy(ctx->side_join);
  if (ctx->table_dev) (void)hipFree(ctx->table_dev);
  for (int k = 0; k < PCV_NUM_STAGES; ++k) {
    if (ctx->stage_b[k]) (void)hipEventDestroy(ctx->stage_b[k]);
    if (ctx->stage_e[k]) (void)hipEventDestroy(ctx->stage_e[k]);
  }
  for (auto& p : ctx->prof_pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (auto& e : ctx->prof_free) (void)hipEventDestroy(e);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* pcv_last_error(const pcv_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

extern "C" int pcv_ctx_synchronize(pcv_ctx* ctx) {
  if (!ctx) return PCV_E_INVALID;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PCV_OK;
}

// Stream hand-off with the caller's runtime (torch, RCCL): order the context's stream after / before another stream
// of the same device without blocking the host. `stream` may be NULL: the legacy default stream (torch's default).
extern "C" int pcv_ctx_wait_stream(pcv_ctx* ctx, void* stream) {
  if (!ctx) return PCV_E_INVALID;
  if ((hipStream_t)stream == ctx->stream) return PCV_OK;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PCV_HIP_CHECK(ctx, hipEventRecord(ctx->xev, (hipStream_t)stream));
  PCV_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->xev, 0));
  return PCV_OK;
}
extern "C" int pcv_ctx_signal_stream(pcv_ctx* ctx, void* stream) {
  if (!ctx) return PCV_E_INVALID;
  if ((hipStream_t)stream == ctx->stream) return PCV_OK;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PCV_HIP_CHECK(ctx, hipEventRecord(ctx->xev, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamWaitEvent((hipStream_t)stream, ctx->xev, 0));
  return PCV_OK;
}

extern "C" int pcv_ctx_trim(pcv_ctx* ctx) {
  if (!ctx) return PCV_E_INVALID;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->pool.trim();
  for (auto& kv : ctx->host_free) (void)hipHostFree(kv.second);
  ctx->host_free.clear();
  return PCV_OK;
}

// ------------------------------------------------------------------------------------------------
// level table (host): reference codec.rs:31-40, node.rs:161, aabb.rs:149-157
// ------------------------------------------------------------------------------------------------
static uint32_t rust_as_u32(double v) {  // Rust `f64 as u32`: NaN -> 0, saturating, truncating
  if (!(v > 0.0)) return 0;
  if (v >= 4294967295.0) return 4294967295u;
  return (uint32_t)v;
}
static int position_encoding(double edge, double resolution) {
  uint32_t min_bits = rust_as_u32(std::log2(edge / resolution)) + 1u;  // wraps like a release build
  if (min_bits <= 8) return PCV_ENC_UINT8;
  if (min_bits <= 16) return PCV_ENC_UINT16;
  if (min_bits <= 24) return PCV_ENC_FLOAT32;
  return PCV_ENC_FLOAT64;
}

int pcv_make_levels(const double bmin[3], const double bmax[3], double resolution, int cap, PcvLevels* lv,
                    int* max_level, std::vector<double>* edges, std::vector<int32_t>* encs) {
  // Cube::bounding: f64::max chain of the extents (aabb.rs:149-157)
  double edge = std::fmax(std::fmax(bmax[0] - bmin[0], bmax[1] - bmin[1]), bmax[2] - bmin[2]);
  std::vector<double> e;
  std::vector<int32_t> c;
  e.push_back(edge);
  c.push_back(position_encoding(edge, resolution));
  int k = 0;
  while (k < cap) {
    ++k;
    edge /= 2.;
    e.push_back(edge);
    c.push_back(position_encoding(edge, resolution));
    if (edge <= resolution) break;  // generation.rs:137: such a node is never split
  }
  if (max_level) *max_level = k;
  if (lv) {
    std::memset(lv, 0, sizeof(*lv));
    for (int a = 0; a < 3; ++a) lv->root_min[a] = bmin[a];
    int nl = k < PCV_MAX_KEY_LEVELS ? k : PCV_MAX_KEY_LEVELS;
    lv->nlevels = nl;
    const int filled = k < PCV_MAX_LEVELS ? k : PCV_MAX_LEVELS;  // the tables cover the deep levels as well
    bool tame = std::fabs(bmin[0]) <= 0x1p+500 && std::fabs(bmin[1]) <= 0x1p+500 && std::fabs(bmin[2]) <= 0x1p+500;
    // largest |coordinate| of any cube min / max in the tree (every cube lies inside the root cube)
    double amax = 0.0;
    for (int a = 0; a < 3; ++a) amax = std::fmax(amax, std::fmax(std::fabs(bmin[a]), std::fabs(bmin[a] + e[0])));
    for (int j = 0; j < PCV_MAX_LEVELS + 2; ++j) lv->digit_half[j] = -1.0;
    for (int j = 0; j <= filled && j < (int)e.size(); ++j) {
      lv->edge[j] = e[j];
      // IEEE division on the host: correctly rounded reciprocal; 0 = "use plain division" (pcv_div_const)
      lv->inv_edge[j] = (e[j] >= 0x1p-100 && e[j] <= 0x1p+100) ? 1.0 / e[j] : 0.0;
      // low word of the double-double reciprocal: (1 - e * yh) is exact in one FMA, divided by e and rounded
      lv->inv_edge_lo[j] = lv->inv_edge[j] != 0.0 ? std::fma(-e[j], lv->inv_edge[j], 1.0) / e[j] : 0.0;
      lv->enc[j] = (uint32_t)c[j];
      tame = tame && lv->inv_edge[j] != 0.0;
      // pcv_digit_from_codes: valid where 1.01 u (2.5 A / e + 3) < 1 / (2 M) (u = 2^-53); required here with a factor
      // of two in hand. Level 0 has no codes (the chain starts from the raw position).
      static const bool digit_shortcut = [] {  // PCV_DIGIT_SHORTCUT=0: always compare against the centre (experiments)
        const char* ev = pcv_experiment("PCV_DIGIT_SHORTCUT");
        return !ev || atoi(ev) != 0;
      }();
      if (digit_shortcut && j >= 1 && (c[j] == PCV_ENC_UINT8 || c[j] == PCV_ENC_UINT16) && std::isfinite(amax) && e[j] > 0.0) {
        const double m = c[j] == PCV_ENC_UINT8 ? 255.0 : 65535.0;
        if ((2.5 * amax / e[j] + 3.0) * 4.04 * m < 0x1p+53) {
          lv->digit_half[j] = c[j] == PCV_ENC_UINT8 ? 127.0 : 32767.0;
          lv->digit_mode[j] = 1;
        }
      }
      // pcv_f32 codes (pcv_chain_dev.h, pcv_bits_from_codes / pcv_f32_code_tie): the same inequality with M = 2^24 — the
      // floats next to 1/2 are 2^-25 away; only the single chain pass looks at digit_mode
      if (digit_shortcut && j >= 1 && c[j] == PCV_ENC_FLOAT32 && std::isfinite(amax) && e[j] > 0.0 &&
          (2.5 * amax / e[j] + 3.0) * 4.04 * 0x1p+24 < 0x1p+53) {
        lv->digit_half[j] = 0.5;
        lv->digit_mode[j] = 2;
      }
    }
    lv->fast_ok = tame ? 1 : 0;
    // "codes from codes" (pcv_chain_dev.h, round 5): the step from the Float32 codes of level j to those of level j + 1.
    // With v the level-j code, b = [v > 1/2] and w = 2 v - b (a float, exactly), the reference's chain computes
    //   t = (RN(RN(fma(v, e_j, m_j)) - RN(m_j + b e_{j+1})) / e_{j+1}) = w + delta,
    //   |delta| <= D = 1.01 (H / e_{j+1} + 3 u),   H = one ulp of the binade of the largest |coordinate| of the root cube
    // (each of the two roundings at that magnitude is off by at most H / 2; the subtraction and the division add at most
    // 2.1 u), and (float)clamp(t) == w whenever thr <= w < 1 for a power of two thr with D < thr 2^-25: the floats next to
    // w are at least thr 2^-24 away. The table stores the high word of the smallest such thr with a factor of two in hand;
    // steps whose thr would exceed 2^-8 are not admitted (most waves would hold a code below it).
    {
      static const bool code_steps = [] {  // PCV_CODE_STEPS=0: every level step in full (experiments)
        const char* ev = pcv_experiment("PCV_CODE_STEPS");
        return !ev || atoi(ev) != 0;
      }();
      const int none = 1 << 20;  // "no such step"
      int cb = none, ce = none;
      if (code_steps && tame && std::isfinite(amax) && amax > 0.0) {
        const double H = std::ldexp(1.0, std::ilogb(amax * (1.0 + 0x1p-40)) - 52);
        for (int j = 1; j + 1 <= filled && j + 1 < (int)e.size() && j <= PCV_MAX_KEY_LEVELS; ++j) {
          if (c[j] != PCV_ENC_FLOAT32 || c[j + 1] != PCV_ENC_FLOAT32 || lv->digit_mode[j] != 2 || !(e[j + 1] > 0.0)) continue;
          const double D = (H / e[j + 1] + 3.0 * 0x1p-53) * 1.01;
          int ex = 0;
          (void)std::frexp(2.0 * D * 0x1p+25, &ex);  // 2 D 2^25 = f 2^ex, 1/2 <= f < 1: thr = 2^ex is strictly above it
          if (ex > -8) continue;
          if (ex < -100) ex = -100;
          lv->code_thr_hi[j] = (uint32_t)(1023 + ex) << 20;
        }
        for (cb = 1; cb <= PCV_MAX_KEY_LEVELS && !lv->code_thr_hi[cb]; ++cb) {
        }
        for (ce = cb; ce <= PCV_MAX_KEY_LEVELS && lv->code_thr_hi[ce]; ++ce) {
        }
        if (cb > PCV_MAX_KEY_LEVELS) cb = ce = none;
        for (int j = ce < PCV_MAX_KEY_LEVELS + 2 ? ce : PCV_MAX_KEY_LEVELS + 2; j < PCV_MAX_KEY_LEVELS + 2; ++j) lv->code_thr_hi[j] = 0;  // one contiguous range
      }
      lv->code_begin = cb;
      lv->code_end = ce;
    }
    {
      const int never = 1 << 20;
      int f16 = never, f8 = never, f32 = never;
      bool monotone = true;
      const int last = filled < (int)e.size() - 1 ? filled : (int)e.size() - 1;
      for (int j = 1; j <= last; ++j) {
        if (c[j] <= PCV_ENC_FLOAT32 && f32 == never) f32 = j;
        if (c[j] <= PCV_ENC_UINT16 && f16 == never) f16 = j;
        if (c[j] == PCV_ENC_UINT8 && f8 == never) f8 = j;
        if (j > 1 && c[j] > c[j - 1]) monotone = false;
      }
      if (f8 != never && f16 == never) f16 = f8;
      lv->first_f32 = monotone ? f32 : never;
      lv->first_u16 = monotone ? f16 : never;
      lv->first_u8 = monotone ? f8 : never;
    }
  }
  if (edges) *edges = e;
  if (encs) *encs = c;
  return PCV_OK;
}

extern "C" int pcv_level_table(const double bbox_min[3], const double bbox_max[3], double resolution, int cap,
                               double* edge, int32_t* encoding) {
  std::vector<double> e;
  std::vector<int32_t> c;
  int ml = 0;
  pcv_make_levels(bbox_min, bbox_max, resolution, cap, nullptr, &ml, &e, &c);
  for (int k = 0; k <= ml; ++k) {
    if (edge) edge[k] = e[k];
    if (encoding) encoding[k] = c[k];
  }
  return ml;
}

extern "C" int pcv_level_shortcuts(const double bbox_min[3], const double bbox_max[3], double resolution, uint32_t* digit_mode,
                                   double* code_threshold) {
  PcvLevels lv;
  int ml = 0;
  pcv_make_levels(bbox_min, bbox_max, resolution, PCV_MAX_LEVELS, &lv, &ml, nullptr, nullptr);
  for (int k = 0; k < PCV_MAX_KEY_LEVELS + 2; ++k) {
    if (digit_mode) digit_mode[k] = lv.digit_mode[k];
    if (code_threshold) {
      double thr = 0.0;
      if (k >= lv.code_begin && k < lv.code_end && lv.code_thr_hi[k]) {
        const uint64_t bits = (uint64_t)lv.code_thr_hi[k] << 32;
        std::memcpy(&thr, &bits, 8);
      }
      code_threshold[k] = thr;
    }
  }
  return ml;
}

// ------------------------------------------------------------------------------------------------
// input staging
// ------------------------------------------------------------------------------------------------
struct DevPoints {
  uint64_t n = 0;
  const double *x = nullptr, *y = nullptr, *z = nullptr;
  PcvRouted routed;  // multi-GPU build: level-1 chain state instead of raw coordinates
  const uint8_t* color = nullptr;
  uint32_t color_stride = 3;
  const float* intensity = nullptr;
};

static int validate_points(pcv_ctx* ctx, const pcv_points* p, bool need_color) {
  if (!p) return ctx->fail(PCV_E_INVALID, "points is null");
  if (p->mem != PCV_MEM_HOST && p->mem != PCV_MEM_DEVICE) return ctx->fail(PCV_E_INVALID, "points.mem must be PCV_MEM_HOST or PCV_MEM_DEVICE");
  if (p->n >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "at most 2^32 - 2 points per call");
  if (p->n > 0 && (!p->x || !p->y || !p->z)) return ctx->fail(PCV_E_INVALID, "x/y/z must be non-null");
  if (need_color) {
    if (p->n > 0 && !p->color) return ctx->fail(PCV_E_INVALID, "color is required (on_disk.rs:20-22: colour is always present)");
    if (p->color_stride != 3 && p->color_stride != 4) return ctx->fail(PCV_E_INVALID, "color_stride must be 3 or 4");
  }
  return PCV_OK;
}

static int stage_points(pcv_ctx* ctx, PcvScratch& sc, const pcv_points* p, bool with_attrs, DevPoints* d) {
  d->n = p->n;
  d->color_stride = p->color_stride;
  if (p->mem == PCV_MEM_DEVICE || p->n == 0) {
    d->x = p->x;
    d->y = p->y;
    d->z = p->z;
    d->color = p->color;
    d->intensity = p->intensity;
    return PCV_OK;
  }
  double *x, *y, *z;
  int rc;
  if ((rc = sc.get(&x, p->n)) || (rc = sc.get(&y, p->n)) || (rc = sc.get(&z, p->n))) return rc;
  if ((rc = ctx->h2d(x, p->x, p->n * 8)) || (rc = ctx->h2d(y, p->y, p->n * 8)) || (rc = ctx->h2d(z, p->z, p->n * 8))) return rc;
  d->x = x;
  d->y = y;
  d->z = z;
  if (with_attrs) {
    uint8_t* c;
    if ((rc = sc.get(&c, p->n * p->color_stride))) return rc;
    if ((rc = ctx->h2d(c, p->color, p->n * p->color_stride))) return rc;
    d->color = c;
    if (p->intensity) {
      float* f;
      if ((rc = sc.get(&f, p->n))) return rc;
      if ((rc = ctx->h2d(f, p->intensity, p->n * 4))) return rc;
      d->intensity = f;
    }
  }
  return PCV_OK;
}

static void host_lap(const char* what, bool reset = false);
// K1 in two halves, so that the caller can do host work (allocations) while the reduction runs.
// The final kernel stores the six doubles straight into the pinned mailbox (host memory the device can write), which holds a
// sentinel until then: the host polls the mailbox instead of waiting for the stream — the blocked wait of a stream synchronize
// wakes up 20-30 us after a 0.4 ms kernel has ended, the poll sees the stores within a few.
static const uint64_t kAabbSentinel = 0x7ff8dead0badbeefull;  // a NaN no fmin / fmax reduction of the kernel can produce
static int device_aabb_launch(pcv_ctx* ctx, PcvScratch& sc, const DevPoints& d) {
  double* partial;
  int rc = sc.get(&partial, (size_t)2048 * 6 + 6);
  if (rc) return rc;
  for (int a = 0; a < 6; ++a) ctx->mailbox[a] = kAabbSentinel;  // nothing queued on the stream writes these slots before the kernel does
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  double* out6 = (double*)ctx->mailbox_dev;
  // the 16-byte vector loads need aligned bases; fall back to staging when the caller's views are not
  if (((uintptr_t)d.x | (uintptr_t)d.y | (uintptr_t)d.z) & 15) {
    double *x, *y, *z;
    if ((rc = sc.get(&x, d.n)) || (rc = sc.get(&y, d.n)) || (rc = sc.get(&z, d.n))) return rc;
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(x, d.x, d.n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(y, d.y, d.n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(z, d.z, d.n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    pcv_launch_aabb(ctx, d.n, x, y, z, partial, out6);
  } else {
    pcv_launch_aabb(ctx, d.n, d.x, d.y, d.z, partial, out6);
  }
  PCV_HIP_CHECK(ctx, hipGetLastError());
  return PCV_OK;
}
static int device_aabb_wait(pcv_ctx* ctx, double bmin[3], double bmax[3]) {
  static const bool poll_on = [] {
    const char* e = pcv_experiment("PCV_AABB_POLL");  // experiments: 0 = wait for the stream
    return !e || atoi(e) != 0;
  }();
  host_lap("", true);
  volatile uint64_t* box = ctx->mailbox;
  bool seen = false;
  if (poll_on) {
    for (uint32_t spin = 0; !seen; ++spin) {
      seen = true;
      for (int a = 0; a < 6; ++a) seen = seen && box[a] != kAabbSentinel;
      // every few thousand reads: is the stream still alive? (a failed launch would never deliver)
      if (!seen && (spin & 0xfff) == 0xfff && hipStreamQuery(ctx->stream) != hipErrorNotReady) break;
    }
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
  }
  if (!seen) PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  host_lap("bbox: wait");
  double h[6];
  std::memcpy(h, ctx->mailbox, sizeof(h));
  for (int a = 0; a < 3; ++a) {
    bmin[a] = h[a];
    bmax[a] = h[3 + a];
  }
  return PCV_OK;
}
static int device_aabb(pcv_ctx* ctx, PcvScratch& sc, const DevPoints& d, double bmin[3], double bmax[3]) {
  if (d.n == 0) {  // Aabb::zero() (generation.rs:269)
    for (int a = 0; a < 3; ++a) bmin[a] = bmax[a] = 0.;
    return PCV_OK;
  }
  int rc = device_aabb_launch(ctx, sc, d);
  return rc ? rc : device_aabb_wait(ctx, bmin, bmax);
}

// ------------------------------------------------------------------------------------------------
// octree object
// ------------------------------------------------------------------------------------------------
// State of a build between pcv_build_begin (through the topology) and pcv_build_finish (encode + promotion).
constexpr int kMailboxResolve = 128;  // u64 slot of the pinned mailbox (pcv_internal.h: 0..63 read-backs, 64..127 replay ranges)
struct PcvBuild {
  pcv_ctx* ctx;
  PcvScratch sc;
  DevPoints d;
  uint64_t n = 0;
  bool stage_times = false;  // PCV_BUILD_STAGE_TIMES of THIS build (a context may have begun another one before the finish)
  PcvLevels lv;
  uint64_t *keys_a = nullptr, *keys_b = nullptr;
  void* sort_scratch = nullptr;
  uint32_t M = 0;
  size_t host_bytes = 0;
  std::vector<uint64_t> pre;  // |pre(node)| stream lengths, bottom-up
  bool deep = false;          // more than PCV_MAX_KEY_LEVELS levels: second key word, prefix_lo in the node table
  int levels = 0;             // levels the level tables are valid for in K5/K6 (full depth, or the deep depth)
  uint64_t* d_prefix_lo = nullptr;
  // single-chain build: the records (true-leaf rank, leaf codes + rgb[, intensity]) already exist when the topology does
  bool spec = false;
  void* spec_payload = nullptr;  // uint4[n]; uint2[n] with 12-byte records
  void* spec_wide = nullptr;     // set: 12-byte records (pcv_internal.h); uint4[n], the codes of Float32-coded leaves
  const uint32_t* spec_rows = nullptr;  // rank counts per sort workgroup (pcv_launch_rank_hist_rows), or null
  PcvSortSecond sort_second;            // the record sort's second pass, held back until the node tables are up (pcv_build_finish)
  uint64_t wide_levels = 0;      // bit k: level k is Float32-coded
  struct FixRange {
    uint32_t lo, count, level;
  };
  std::vector<FixRange> fix_ranges;  // sorted slots whose points replay the chain after the record sort
  const uint32_t* spec_map_dev = nullptr;  // set: the record sort's first upsweep applies the rank map
  uint32_t spec_map_entries = 0;
  // true leaves below a split first candidate (PcvTrueTree::cont_nodes / cont_from): their chain is continued after the sort
  std::vector<uint32_t> cont_nodes, cont_from;
  // the record sort (queue_record_sort): buffers and where the sorted records ended up
  bool sort_queued = false, rec_in_a = true;
  void *pay_a = nullptr, *pay_b = nullptr;
  // device resolve (spec_resolve_kernel drives the record sort) vs host resolve (pcv_spec_resolve builds the tables): what the
  // device reported — {true leaves, too shallow} — lands in mailbox[kMailboxResolve] and pcv_build_finish compares it with
  // the host's tree before anything of the build is handed out; check_map: the whole rank map is compared too
  bool resolve_on_device = false, resolve_check_map = false;
  uint32_t resolve_host_leaves = 0;
  std::vector<uint32_t> resolve_host_map;
  std::vector<uint8_t> resolve_host_inner;  // T'' node is an inner node: its map entry is never read (the device leaves it unwritten)
  PcvSortPayload pl;
  bool color_late = false;  // single-chain build: the chain pass wrote its 12-byte records without the colour
  explicit PcvBuild(pcv_ctx* c) : ctx(c), sc(c) {}
  // A build dropped between begin and finish (an error in finish before the held-back pass was queued, a tree freed without
  // finish): the pass's layout kernels on the side stream write into `sc`'s sort scratch; the main stream must be ordered behind
  // them before the members below hand that scratch back to the pool
  ~PcvBuild() {
    if (sort_second.pending && sort_second.join_side) (void)ctx->side_end();
  }
};

extern "C" void pcv_octree_free(pcv_octree* t) {
  if (!t) return;
  delete t->pending;
  if (t->ctx) {
    pcv_octree_release_query(t);
    t->ctx->host_release(t->h_xyz.p);
    t->ctx->host_release(t->h_rgb.p);
    t->ctx->host_release(t->h_int.p);
    t->ctx->dev_free(t->d_xyz);
    t->ctx->dev_free(t->d_rgb);
    t->ctx->dev_free(t->d_int);
  }
  delete t;
}
extern "C" uint64_t pcv_octree_num_nodes(const pcv_octree* t) { return t ? t->nodes.size() : 0; }
extern "C" uint64_t pcv_octree_num_points(const pcv_octree* t) { return t ? t->num_points : 0; }
extern "C" int pcv_octree_has_intensity(const pcv_octree* t) { return t && t->has_intensity; }
extern "C" int pcv_octree_node(const pcv_octree* t, uint64_t i, pcv_node_info* out) {
  if (!t || !out || i >= t->nodes.size()) return PCV_E_INVALID;
  *out = t->nodes[i];
  return PCV_OK;
}
extern "C" void pcv_octree_meta(const pcv_octree* t, double* resolution, double bbox_min[3], double bbox_max[3],
                                int* version) {
  if (!t) return;
  if (resolution) *resolution = t->resolution;
  for (int a = 0; a < 3; ++a) {
    if (bbox_min) bbox_min[a] = t->bbox_min[a];
    if (bbox_max) bbox_max[a] = t->bbox_max[a];
  }
  if (version) *version = 13;  // CURRENT_VERSION, reference src/lib.rs:48
}
extern "C" int pcv_octree_stage_ms(const pcv_octree* t, float* ms, int cap) {
  if (!t || !ms) return 0;
  int n = cap < PCV_NUM_STAGES ? cap : PCV_NUM_STAGES;
  for (int i = 0; i < n; ++i) ms[i] = t->stage_ms[i];
  return n;
}
extern "C" void pcv_octree_build_info(const pcv_octree* t, int* key_levels, int* attempts) {
  if (!t) return;
  if (key_levels) *key_levels = t->key_levels;
  if (attempts) *attempts = t->key_attempts;
}
extern "C" void pcv_octree_spec_stats(const pcv_octree* t, uint64_t stats[4]) {
  if (!t || !stats) return;
  for (int k = 0; k < 4; ++k) stats[k] = t->spec_stats[k];
}
extern "C" int pcv_octree_record_bytes(const pcv_octree* t) { return t ? t->record_bytes : 0; }
extern "C" uint64_t pcv_octree_spec_continued(const pcv_octree* t) { return t ? t->spec_continued : 0; }
extern "C" uint64_t pcv_octree_wide_pool_entries(const pcv_octree* t) { return t ? t->wide_pool_entries : 0; }
extern "C" uint64_t pcv_octree_settled_in_sort(const pcv_octree* t) { return t ? t->settled_in_sort : 0; }
extern "C" int pcv_octree_device_blob(const pcv_octree* t, int which, const void** dptr, uint64_t* len) {
  if (!t || !dptr || !len || which < 0 || which > 2) return PCV_E_INVALID;
  *dptr = which == 0 ? t->d_xyz : (which == 1 ? t->d_rgb : t->d_int);
  *len = which == 0 ? t->xyz_bytes : (which == 1 ? t->rgb_bytes : t->int_bytes);
  return PCV_OK;
}

int pcv_octree_fetch_host(pcv_octree* t) {
  if (t->host_valid) return PCV_OK;
  pcv_ctx* ctx = t->ctx;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int hrc;
  if (t->xyz_bytes && (hrc = ctx->host_alloc((void**)&t->h_xyz.p, t->xyz_bytes))) return hrc;
  if (t->rgb_bytes && (hrc = ctx->host_alloc((void**)&t->h_rgb.p, t->rgb_bytes))) return hrc;
  if (t->int_bytes && (hrc = ctx->host_alloc((void**)&t->h_int.p, t->int_bytes))) return hrc;
  if (t->xyz_bytes) PCV_HIP_CHECK(ctx, hipMemcpyAsync(t->h_xyz.data(), t->d_xyz, t->xyz_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (t->rgb_bytes) PCV_HIP_CHECK(ctx, hipMemcpyAsync(t->h_rgb.data(), t->d_rgb, t->rgb_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (t->int_bytes) PCV_HIP_CHECK(ctx, hipMemcpyAsync(t->h_int.data(), t->d_int, t->int_bytes, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  t->host_valid = true;
  return PCV_OK;
}

int pcv_bytes_per_coordinate(uint32_t enc) { return enc == PCV_ENC_UINT8 ? 1 : enc == PCV_ENC_UINT16 ? 2 : enc == PCV_ENC_FLOAT32 ? 4 : 8; }

extern "C" int pcv_octree_node_data(pcv_octree* t, uint64_t i, int which, const uint8_t** data, uint64_t* len) {
  if (!t || !data || !len || i >= t->nodes.size() || which < 0 || which > 2) return PCV_E_INVALID;
  if (!t->directory.empty()) return pcv_octree_read_node_file(t, i, which, data, len);
  int rc = pcv_octree_fetch_host(t);
  if (rc) return rc;
  const pcv_node_info& nd = t->nodes[i];
  uint64_t np = (uint64_t)nd.num_points;
  if (which == 0) {
    *data = t->h_xyz.data() + nd.xyz_offset;
    *len = np * 3 * (uint64_t)pcv_bytes_per_coordinate(nd.encoding);
  } else if (which == 1) {
    *data = t->h_rgb.data() + nd.point_offset * 3;
    *len = np * 3;
  } else {
    if (!t->has_intensity) {
      *data = nullptr;
      *len = 0;
    } else {
      *data = t->h_int.data() + nd.point_offset * 4;
      *len = np * 4;
    }
  }
  return PCV_OK;
}

extern "C" int pcv_octree_copy_node(const pcv_octree* t, uint64_t i, int which, void* dst, uint64_t capacity, int mem) {
  if (!t || i >= t->nodes.size() || which < 0 || which > 2 || (mem != PCV_MEM_HOST && mem != PCV_MEM_DEVICE)) return PCV_E_INVALID;
  pcv_ctx* ctx = t->ctx;
  if (!t->directory.empty()) return ctx->fail(PCV_E_INVALID, "pcv_octree_copy_node works on built octrees (device blobs)");
  const pcv_node_info& nd = t->nodes[i];
  const uint64_t np = (uint64_t)nd.num_points;
  const uint8_t* src = nullptr;
  uint64_t len = 0;
  if (which == 0) {
    src = t->d_xyz + nd.xyz_offset;
    len = np * 3 * (uint64_t)pcv_bytes_per_coordinate(nd.encoding);
  } else if (which == 1) {
    src = t->d_rgb + nd.point_offset * 3;
    len = np * 3;
  } else if (t->has_intensity) {
    src = t->d_int + nd.point_offset * 4;
    len = np * 4;
  }
  if (len > capacity) return ctx->fail(PCV_E_INVALID, "destination too small for the node's bytes");
  if (len == 0) return PCV_OK;
  if (!dst) return ctx->fail(PCV_E_INVALID, "dst is null");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, len, mem == PCV_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
  // device destinations stay asynchronous on the context's stream (pcv_ctx_synchronize, or stream order, completes them)
  if (mem == PCV_MEM_HOST) PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PCV_OK;
}

// Batch form of pcv_octree_copy_node for the multi-GPU top merge: every rank copies its (sparsely filled, global-size)
// root / level-1 nodes into one buffer that is then all-reduced — one call instead of one per node and file kind.
extern "C" int pcv_octree_copy_nodes(const pcv_octree* t, const pcv_node_copy* copies, uint64_t count, void* dst, uint64_t capacity,
                                     int mem) {
  if (!t || (count && !copies) || (mem != PCV_MEM_HOST && mem != PCV_MEM_DEVICE)) return PCV_E_INVALID;
  pcv_ctx* ctx = t->ctx;
  if (!t->directory.empty()) return ctx->fail(PCV_E_INVALID, "pcv_octree_copy_nodes works on built octrees (device blobs)");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const hipMemcpyKind kind = mem == PCV_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  for (uint64_t k = 0; k < count; ++k) {
    const pcv_node_copy& c = copies[k];
    if (c.node >= t->nodes.size()) return ctx->fail(PCV_E_INVALID, "pcv_octree_copy_nodes: node index out of range");
    const pcv_node_info& nd = t->nodes[c.node];
    const uint64_t np = (uint64_t)nd.num_points;
    const uint8_t* src[3] = {t->d_xyz + nd.xyz_offset, t->d_rgb + nd.point_offset * 3, t->has_intensity ? t->d_int + nd.point_offset * 4 : nullptr};
    const uint64_t len[3] = {np * 3 * (uint64_t)pcv_bytes_per_coordinate(nd.encoding), np * 3, t->has_intensity ? np * 4 : 0};
    for (int w = 0; w < 3; ++w) {
      if (c.dst_offset[w] == UINT64_MAX || len[w] == 0) continue;
      if (c.dst_offset[w] > capacity || len[w] > capacity - c.dst_offset[w])
        return ctx->fail(PCV_E_INVALID, "pcv_octree_copy_nodes: destination too small for a node's bytes");
      if (!dst) return ctx->fail(PCV_E_INVALID, "dst is null");
      PCV_HIP_CHECK(ctx, hipMemcpyAsync((uint8_t*)dst + c.dst_offset[w], src[w], len[w], kind, ctx->stream));
    }
  }
  if (mem == PCV_MEM_HOST) PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PCV_OK;
}

// ------------------------------------------------------------------------------------------------
// the build
// ------------------------------------------------------------------------------------------------
static uint64_t ceil8(uint64_t v) { return (v + 7) / 8; }

static int build_begin_impl(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points,
                            const pcv_routed_points* routed, pcv_octree** out, const PcvTrueTree* given_tree = nullptr);
extern "C" int pcv_build_begin(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points,
                               pcv_octree** out) {
  if (!ctx) return PCV_E_INVALID;
  if (!out) return ctx->fail(PCV_E_INVALID, "out is null");
  *out = nullptr;
  if (!params) return ctx->fail(PCV_E_INVALID, "params is null");
  int rc = validate_points(ctx, points, true);
  if (rc) return rc;
  return build_begin_impl(ctx, params, points, nullptr, out);
}
extern "C" int pcv_build_begin_routed(pcv_ctx* ctx, const pcv_build_params* params, const pcv_routed_points* routed,
                                      pcv_octree** out) {
  if (!ctx) return PCV_E_INVALID;
  if (!out) return ctx->fail(PCV_E_INVALID, "out is null");
  *out = nullptr;
  if (!params || !routed) return ctx->fail(PCV_E_INVALID, "null argument");
  if (routed->n >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "at most 2^32 - 2 points per call");
  if (routed->n > 0 && (!routed->oct_rgb || !routed->cx || !routed->cy || !routed->cz))
    return ctx->fail(PCV_E_INVALID, "cx, cy, cz and oct_rgb must be non-null");
  if (params->flags & PCV_BUILD_COMPUTE_BBOX) return ctx->fail(PCV_E_INVALID, "routed points need the global bounding box");
  return build_begin_impl(ctx, params, nullptr, routed, out);
}
// The single-chain topology (pcv_spec.h): sample -> predicted tree T'' -> ONE chain pass for all points -> exact counts
// per predicted leaf -> true tree on the host. On success (*used) the records (true-leaf rank in the first half of
// keys_a, payload in bs->spec_payload, intensity bits in the second half of keys_a) are what K5 would have produced and
// `tt` is the node table K4 would have produced. *used == false: the prediction did not cover the tree (or the sample
// says the tree is deeper than one key word): the caller runs the exact pipeline; nothing of this attempt is kept.
// PCV_HOST_TIMING=1: host-side lap times of the single-chain build's critical section (counts on the host -> first
// sort kernel queued), printed to stderr
#include <atomic>
#include <chrono>
static void host_lap(const char* what, bool reset) {
  static const bool on = pcv_experiment("PCV_HOST_TIMING") != nullptr;
  static std::chrono::steady_clock::time_point t0;
  if (!on) return;
  const auto now = std::chrono::steady_clock::now();
  if (!reset) fprintf(stderr, "[host] %-24s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t0).count());
  t0 = now;
}

// The record sort's second pass settles the leaves' points itself (PcvSortFuse) where it can: the pass is held back until the
// node tables are on the device. PCV_SETTLE_IN_SORT=0 (libpcv_hip_exp.so): the sort runs to its end, `settle` reads the records.
static bool pcv_settle_in_sort() {
  static const bool on = [] {
    const char* e = pcv_experiment("PCV_SETTLE_IN_SORT");
    const char* l = pcv_experiment("PCV_SETTLE_BY_LEAF");
    return (!e || atoi(e) != 0) && (!l || atoi(l) != 0) && pcv_climb16_enabled();
  }();
  return on;
}

// K5 (exact pipeline only: `wt` set) + K3 stable record sort by leaf rank, queued on the stream; the outcome is left in
// the build state for K6. num_leaves only sizes the digits: any upper bound of the number of true leaves will do.
// record = rank (u32) + one 16-byte payload {code x, code y, code z, rgba}; optional 4-byte planes for the intensity
// and, when some leaf level is Float64-encoded, the high words of the codes. The key buffers are dead by now: each
// (8n bytes) hosts one rank array; payloads get their own buffers.
static int queue_record_sort(pcv_ctx* ctx, PcvBuild* bs, pcv_octree* t, const PcvWalkTables* wt, uint32_t num_leaves, bool wide) {
  PcvScratch& sc = bs->sc;
  DevPoints& d = bs->d;
  PcvLevels& lv = bs->lv;
  const uint64_t n = bs->n;
  int rc;
  uint32_t* rank_a = (uint32_t*)bs->keys_a;
  uint32_t* rank_b = (uint32_t*)bs->keys_b;
  const bool compact = bs->spec_wide != nullptr;  // 12-byte records: uint2 payloads
  uint4 *pay_a = (uint4*)bs->spec_payload, *pay_b;
  if (compact) {
    uint2* b2;
    if ((rc = sc.get(&b2, n))) return rc;
    pay_b = (uint4*)b2;
  } else if ((!pay_a && (rc = sc.get(&pay_a, n))) || (rc = sc.get(&pay_b, n))) {
    return rc;
  }
  PcvSortPayload& pl = bs->pl;
  pl = PcvSortPayload();
  pl.vec_in = pay_a;
  pl.vec_out = pay_b;
  pl.vec_bytes = compact ? 8 : 16;
  t->record_bytes = compact ? 12 : 20;
  pl.nwords = (t->has_intensity ? 1 : 0) + (wide ? 3 : 0);
  for (int w = 0; w < pl.nwords; ++w) {
    if (w == 0) {  // first plane fits in the second half of the key buffers
      pl.in[0] = (uint32_t*)bs->keys_a + n;
      pl.out[0] = (uint32_t*)bs->keys_b + n;
    } else if ((rc = sc.get(&pl.in[w], n)) || (rc = sc.get(&pl.out[w], n))) {
      return rc;
    }
  }
  if (bs->spec && t->has_intensity) pl.first_in0 = reinterpret_cast<const uint32_t*>(d.intensity);  // float bits, record order
  const int w_int = t->has_intensity ? 0 : -1;
  const int w_hi = wide ? (t->has_intensity ? 1 : 0) : -1;
  if (!bs->spec) {  // the single-chain build wrote (true-leaf rank, leaf codes, rgb[, intensity]) while it found the topology
    ctx->stage_begin(PCV_STAGE_LEAF_ENCODE);
    pcv_launch_leaf_encode(ctx, lv, *wt, n, d.x, d.y, d.z, d.routed, d.color, d.color_stride, d.intensity, rank_a, pay_a,
                           wide ? pl.in[w_hi] : nullptr, wide ? pl.in[w_hi + 1] : nullptr, wide ? pl.in[w_hi + 2] : nullptr,
                           w_int >= 0 ? pl.in[w_int] : nullptr);
    ctx->stage_end(PCV_STAGE_LEAF_ENCODE);
  }
  ctx->stage_begin(PCV_STAGE_SORT_RECORDS);
  int rank_bits = 1;
  while ((1ull << rank_bits) < num_leaves) ++rank_bits;
  bool rec_in_a = true;
  if (bs->spec && bs->color_late) {
    // the first pass of the rows form reads the colour as it loads the records; any other form of the sort (the rank-count rows
    // did not fit, experiments) gets it joined in place first
    const bool first_pass_joins = bs->spec_map_dev && compact && pl.nwords <= 1 && bs->spec_rows && pcv_sort_first_pass_joins_color(n);
    if (first_pass_joins) {
      pl.color_in = d.color;
      pl.color_stride = d.color_stride;
    } else {
      pcv_launch_join_color(ctx, n, d.color, d.color_stride, rank_a, pay_a);
    }
  }
  if (bs->spec_map_dev)
    rc = pcv_radix_sort_records_mapped(ctx, rank_a, rank_b, n, rank_bits, &pl, bs->sort_scratch, bs->spec_map_dev,
                                       bs->spec_map_entries, &rec_in_a, compact && pl.nwords <= 1 ? bs->spec_rows : nullptr,
                                       compact && pl.nwords <= 1 && !wide && pcv_settle_in_sort() ? &bs->sort_second : nullptr);
  else
    rc = pcv_radix_sort_u32(ctx, rank_a, rank_b, n, 0, rank_bits, &pl, bs->sort_scratch, &rec_in_a);
  if (rc) return rc;
  ctx->stage_end(PCV_STAGE_SORT_RECORDS);
  bs->sort_queued = true;
  bs->rec_in_a = rec_in_a;
  bs->pay_a = pay_a;
  bs->pay_b = pay_b;
  return PCV_OK;
}

// single-chain build: the few leaves whose records hold no usable codes (bs->fix_ranges, PCV_SPEC_MAP_REPLAY) replay their
// chain from the coordinates now that the record sort has made them contiguous; queued behind the sort.
static int queue_replay(pcv_ctx* ctx, PcvBuild* bs) {
  if (!bs->spec || bs->fix_ranges.empty()) return PCV_OK;
  hipStream_t st = ctx->stream;
  DevPoints& d = bs->d;
  int rc;
  const void* s_pay = bs->rec_in_a ? (const void*)bs->pay_a : (const void*)bs->pay_b;
  const uint32_t nr = (uint32_t)bs->fix_ranges.size();
  uint32_t* d_ranges;
  if ((rc = bs->sc.get(&d_ranges, (size_t)nr * 4 + 4))) return rc;
  // staging: the second half of the pinned mailbox block is reserved for these ranges (32 of them) — the upload is
  // queued behind the record sort and the caller gets control back before it has run, so the slot must not be one that
  // other entry points of the context write (they use the first half); more ranges (tiny capacities in tests) wait
  // for the queued work and take the big block
  uint32_t* h_ranges = (uint32_t*)(ctx->mailbox + 64);
  if (nr > 32) {
    PCV_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if ((rc = ctx->pinned_spec_reserve((size_t)nr * 16 + 64))) return rc;
    h_ranges = (uint32_t*)ctx->pinned_spec;
  }
  uint32_t before = 0;
  for (uint32_t k = 0; k < nr; ++k) {
    h_ranges[4 * k + 0] = bs->fix_ranges[k].lo;
    h_ranges[4 * k + 1] = before;
    h_ranges[4 * k + 2] = bs->fix_ranges[k].level;
    h_ranges[4 * k + 3] = 0;
    before += bs->fix_ranges[k].count;
  }
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(d_ranges, h_ranges, (size_t)nr * 16, hipMemcpyHostToDevice, st));
  pcv_launch_spec_replay(ctx, bs->lv, d_ranges, nr, before, d.x, d.y, d.z, d.routed, (void*)s_pay, bs->spec_wide,
                         (uint32_t)pcv_pool_region_entries(bs->n));
  return PCV_OK;
}

static int single_chain_topology(pcv_ctx* ctx, PcvBuild* bs, pcv_octree* t, const pcv_build_params* params,
                                 uint32_t max_points, int full_levels, PcvNodeTableDev& nt, PcvTrueTree* tt, bool* used) {
  *used = false;
  hipStream_t st = ctx->stream;
  PcvScratch& sc = bs->sc;
  DevPoints& d = bs->d;
  PcvLevels& lv = bs->lv;
  const uint64_t n = bs->n;
  int rc;
  // sample stride: every 64th point (>= 1 500 sample points per full node at the reference's capacity: a candidate band of
  // +-13 % around it); small forced builds (tests) sample more densely. Round 5, one call, 100 M points: stride 32 / 48 / 64 /
  // 96 -> 5.02 / 4.97 / 4.95-4.98 / 4.96 ms per build — the sample phase shrinks by 0.07 ms, 1.5 M more points continue
  // their chain in the settle pass, the prediction holds either way (profiles/r05_ab_sample_stride.json)
  uint64_t stride = 64;
  if (const char* e = pcv_experiment("PCV_SPEC_STRIDE")) stride = (uint64_t)std::max(1, atoi(e));  // experiments
  while (stride > 1 && n / stride < 4096) stride >>= 1;
  // a node at the capacity must still hold a few dozen sample points, or the band around the capacity (five standard
  // deviations of the scaled count) swallows every node and the prediction opens all of them (tiny capacities in tests)
  while (stride > 1 && (uint64_t)max_points / stride < 64) stride >>= 1;
  const uint64_t ns = n / stride;
  if (ns == 0) return PCV_OK;
  PcvSpecParams sp;
  sp.cap = max_points;
  sp.resolution = params->resolution;
  sp.edge = lv.edge;
  sp.nlevels = full_levels;
  sp.force_mask = (params->flags >> 8) & 0xffu;
  sp.scale = (double)stride;
  // band of five standard deviations of the scaled sample count of a node that holds exactly `cap` points
  sp.delta = stride == 1 ? 0.0 : std::fmin(0.9, std::fmax(0.02, 5.0 * std::sqrt((double)stride / (double)max_points)));

  ctx->stage_begin(PCV_STAGE_CHAIN_KEYS);
  // The sample keys cover `sample_levels` levels first (below); a sample tree that wants to go deeper is keyed again at
  // full depth.
  // The predicted tree is built on the device (sample split -> spec_tree kernels): the one chain pass starts without a
  // host round trip, and the host mirrors the tree (one small asynchronous copy) while that pass runs.
  constexpr uint32_t kFirst = 16384;  // T'' nodes mirrored by the first copy (a 100 M-point tree has ~7 500)
  // T'' nodes at most (at least 4 096 records: the chain pass may mirror the table's first 2 048 in LDS without asking how many exist)
  const size_t tcap = std::max<size_t>(1 + 8 * (size_t)nt.capacity, 4096);
  uint32_t *d_ord, *d_walk, *d_sparent, *d_info, *d_counts, *d_map, *d_pool_ctr;
  uint8_t* d_slevel;
  if ((rc = sc.get(&d_ord, nt.capacity)) || (rc = sc.get(&d_walk, tcap)) || (rc = sc.get(&d_sparent, tcap)) ||
      (rc = sc.get(&d_slevel, tcap)) || (rc = sc.get(&d_info, 64)) || (rc = sc.get(&d_pool_ctr, kPcvPoolRegions + tcap + 4)) ||
      (rc = sc.get(&d_map, tcap)))
    return rc;
  d_counts = d_pool_ctr + kPcvPoolRegions;  // the pool counters and the exact counts travel to the host in ONE copy
  uint32_t* rank = (uint32_t*)bs->keys_a;
  uint4* payload;
  // 12-byte records (pcv_internal.h) unless switched off (PCV_COMPACT_RECORDS=0, experiments) or the predicted tree could
  // outgrow the 24 rank bits of the key
  static const bool compact_on = [] {
    const char* e = pcv_experiment("PCV_COMPACT_RECORDS");
    return !e || atoi(e) != 0;
  }();
  // ... or the pool of Float32 codes (kPcvPoolRegions regions, pcv_internal.h) could outgrow 32-bit entry numbers
  const uint64_t pool_cap = pcv_pool_region_entries(n), pool_entries = pool_cap * kPcvPoolRegions;
  const bool compact = compact_on && tcap <= (1u << 24) && pool_entries <= 0xffffffffull;
  uint4* wide = nullptr;
  uint64_t wide_levels = 0;
  for (int k = 0; k <= full_levels && k < 64; ++k)
    if (lv.enc[k] > PCV_ENC_UINT16) wide_levels |= 1ull << k;
  if (compact) {
    uint2* p2;
    if ((rc = sc.get(&p2, n)) || (rc = sc.get(&wide, pool_entries))) return rc;
    payload = (uint4*)p2;
  } else if ((rc = sc.get(&payload, n))) {
    return rc;
  }
  // the intensity plane needs no copy: the chain pass leaves its records in input order, so the record sort's first pass
  // reads the caller's own array (PcvSortPayload::first_in0, queue_record_sort) — 8 bytes per point less in the chain pass
  uint32_t* inten_bits = nullptr;
  uint8_t* depth_grid = nullptr;
  if (n >= (1u << 20) && (rc = sc.get(&depth_grid, pcv_spec_depth_grid_bytes()))) return rc;  // small builds: not worth a 2 MiB fill
  const size_t h_walk = 256, h_parent = h_walk + (size_t)kFirst * 4, h_level = h_parent + (size_t)kFirst * 4;
  const size_t h_first_bytes = h_level + kFirst;
  if ((rc = ctx->pinned_spec_reserve(h_first_bytes + 256))) return rc;
  const double upper = (double)sp.cap * (1.0 + sp.delta) / sp.scale;
  PcvSpecTree tree;
  uint32_t info[4] = {0, 0, 0, 0};
  uint64_t* small_partner = nullptr;
#ifdef PCV_EXPERIMENTS
  uint32_t* d_sample_counts = nullptr;
#endif
  // Levels the sample keys cover first: a uniform cloud reaches the capacity at level log8(n / capacity); clustered clouds
  // go deeper, so eight levels on top (12 levels for the 100 M bench cloud whose deepest leaf sits at level 10, 13 for
  // 1 B points), at most 14. Every level less is a tenth of the sample's chain, three bits of its key sort and two
  // launches of its split; a sample tree that wants to go deeper is keyed again at full depth (below).
  int sample_levels;
  {
    int uniform = 0;
    for (uint64_t per_node = n / (uint64_t)max_points + 1; per_node > 1; per_node = (per_node + 7) / 8) ++uniform;
    sample_levels = std::min(14, std::max(10, uniform + 8));
    if (const char* e = pcv_experiment("PCV_SAMPLE_LEVELS")) sample_levels = std::max(1, atoi(e));
    if (sample_levels > full_levels) sample_levels = full_levels;
  }
  for (;;) {
    lv.nlevels = sample_levels;
    // keys_a doubles as the rank array of the chain pass below: a second round must not start before the first
    // round's pass is done with it — same stream, so it is ordered
    uint64_t* skeys_a = bs->keys_b;  // the sample keys (and their sort partner) live in keys_b: 2 x n / 32 x 8 B of its 8 n
    const uint64_t soff = ((ns + 31) & ~31ull) + 32;  // keeps the partner 256-byte aligned
    uint64_t* skeys_b = bs->keys_b + soff;
    if (soff + ns > n) {  // tiny forced builds sample every point: the partner gets its own buffer
      if (!small_partner && (rc = sc.get(&small_partner, ns + 32))) return rc;
      skeys_b = small_partner;
    }
    // The sample is taken in clumps of 8 consecutive points (one clump every 8 x stride): read one by one, every
    // sampled coordinate costs a full cache line (1.2 GB and 0.23 ms for the 3.1 M sample points of a 100 M cloud)
    static const uint32_t clump_shift = [] {
      const char* e = pcv_experiment("PCV_SAMPLE_CLUMP_SHIFT");  // experiments: 0 = single points
      return e ? (uint32_t)std::min(6, std::max(0, atoi(e))) : 3u;
    }();
    uint32_t* one = nullptr;  // (scratch of the one-launch-per-digit key sort: libpcv_hip_exp.so, PCV_SAMPLE_ONESWEEP=1; pcv_sort.hip)
    size_t one_zero_words = 0;
#ifdef PCV_EXPERIMENTS
    static const bool onesweep_on = [] {
      const char* e = pcv_experiment("PCV_SAMPLE_ONESWEEP");
      return e && atoi(e) != 0;
    }();
    const int sbits = 3 * sample_levels;
    if (onesweep_on && pcv_onesweep_fits(ns, sbits)) {
      if ((rc = sc.get(&one, pcv_onesweep_scratch_words(ns, sbits) + 4))) return rc;
      one_zero_words = pcv_onesweep_zero_words(ns, sbits);
    }
#endif
    host_lap("bbox -> sample keys");
    pcv_launch_chain_keys(ctx, lv, ns, stride, d.x, d.y, d.z, skeys_a, false, d.routed, stride > 1 ? clump_shift : 0u, one, one_zero_words);
    bool in_a = true;
    host_lap("", true);
    // PCV_SAMPLE_COUNTS=1 (libpcv_hip_exp.so only): the sample tree by COUNTING the keys, three levels per launch pair
    // (pcv_topology.hip): 9 launches instead of 27 and no sort — and 2-3 x SLOWER, measured: the 1.5 M keys cost 4.7 M
    // device-scope atomics per group of levels and this part retires ~9 G of those per second (0.54 + 0.38 ms for the two
    // middle groups against 0.28 ms for the whole key sort + split; profiles/r05_ab_sample_tree_by_counting_dropped.json)
    const uint32_t thr_s = pcv_spec_sample_threshold(sp);
    bool counted = false;
#ifdef PCV_EXPERIMENTS
    static const bool counts_on = [] {
      const char* e = pcv_experiment("PCV_SAMPLE_COUNTS");
      return e && atoi(e) != 0;
    }();
    if (counts_on && thr_s > 0 && nt.max_open >= ns / thr_s + 16 && sample_levels <= PCV_MAX_KEY_LEVELS) {
      if (!d_sample_counts && (rc = sc.get(&d_sample_counts, pcv_sample_count_scratch_words(nt.capacity, nt.max_open, full_levels)))) return rc;
      // (a count only matters up to the larger of the split threshold and the candidate band's upper end)
      const double sat_d = std::fmax((double)thr_s, std::ceil(upper)) + 2.0;
      const uint32_t sat = sat_d >= 4294967000.0 ? 0xfffffff0u : (uint32_t)sat_d;
      pcv_launch_sample_tree_counts(ctx, nt, skeys_a, (uint32_t)ns, lv, params->resolution, thr_s, sp.force_mask, d_sample_counts, sat);
      host_lap("sample tree (counting) queued");
      counted = true;
    }
#endif
    if (!counted) {
#ifdef PCV_EXPERIMENTS
    if (one) rc = pcv_sort_keys_onesweep(ctx, skeys_a, skeys_b, ns, 3 * (PCV_MAX_KEY_LEVELS - sample_levels), 3 * PCV_MAX_KEY_LEVELS, one, &in_a);
    else
#endif
      rc = pcv_radix_sort_u64(ctx, skeys_a, skeys_b, ns, 3 * (PCV_MAX_KEY_LEVELS - sample_levels), 3 * PCV_MAX_KEY_LEVELS, nullptr,
                                 bs->sort_scratch, &in_a);
    if (rc) return rc;
    host_lap("sample sort queued");
    pcv_launch_node_split(ctx, nt, in_a ? skeys_a : skeys_b, false, (uint32_t)ns, lv, params->resolution, thr_s, sp.force_mask);
    host_lap("sample split queued");
    }
    pcv_launch_spec_tree(ctx, nt, upper, sp.force_mask, d_ord, d_walk, d_sparent, d_slevel, d_info, d_pool_ctr);
    host_lap("spec tree queued");
    uint8_t* hs = (uint8_t*)ctx->pinned_spec;
    const size_t first = tcap < kFirst ? tcap : kFirst;
    // The host's mirror of the tree and the zeroed counters travel on the side stream: on `stream` each of these small
    // operations would sit (with its ~12 us hand-over) between the tree kernels and the chain pass. The fork point is
    // recorded now, the chain pass is queued next (the GPU has caught up with the host by here, so every API call
    // before that launch is idle GPU time), and only then is the side stream fed.
    PCV_HIP_CHECK(ctx, hipEventRecord(ctx->side_fork, st));
    ctx->stage_end(PCV_STAGE_CHAIN_KEYS);

    // ---- the one chain pass (queued before the host has seen the tree) ----
    ctx->stage_begin(PCV_STAGE_LEAF_ENCODE);
    lv.nlevels = full_levels;
    // round 6, measured and NOT shipped (PCV_COLOR_LATE=1, libpcv_hip_exp.so; profiles/r06_ab_colour_joins_in_the_sort_dropped.json):
    // the 12-byte records leave the pass without their colour and the record sort's first pass — which reads every record in
    // input order anyway — fetches it from the caller's array. The chain pass gains 0.13 ms (1.86 -> 1.73 at 100 M points), the
    // sort's first pass loses 0.31 (0.59 -> 0.90): eight more (unaligned) loads per lane on top of its sixteen — that pass is
    // bound by the issue of its vector-memory instructions, not by bytes. Same octree either way.
    static const bool color_late_on = [] {
      const char* e = pcv_experiment("PCV_COLOR_LATE");
      return e && atoi(e) != 0;
    }();
    bs->color_late = color_late_on && compact;
    // the exact counters (d_counts) are cleared by the depth-grid kernel in front of the pass where there is one; small builds
    // clear them with a fill on `stream`
    const size_t counts_words = (tcap + 3) & ~(size_t)3;
    const bool zero_in_grid = depth_grid && (((uintptr_t)d_counts & 15) == 0);
    if (!zero_in_grid) PCV_HIP_CHECK(ctx, hipMemsetAsync(d_counts, 0, tcap * 4, st));
    pcv_launch_spec_encode(ctx, lv, d_walk, n, d.x, d.y, d.z, d.routed, d.color, d.color_stride, d.intensity, rank, payload,
                           inten_bits, depth_grid, wide, d_pool_ctr /* zeroed by the spec_tree kernels */, d_info, bs->color_late,
                           zero_in_grid ? d_counts : nullptr, zero_in_grid ? counts_words : 0);
    ctx->stage_end(PCV_STAGE_LEAF_ENCODE);
    host_lap("chain pass queued");

    PCV_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->side, ctx->side_fork, 0));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(hs, d_info, 16, hipMemcpyDeviceToHost, ctx->side));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(hs + h_walk, d_walk, first * 4, hipMemcpyDeviceToHost, ctx->side));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(hs + h_parent, d_sparent, first * 4, hipMemcpyDeviceToHost, ctx->side));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(hs + h_level, d_slevel, first, hipMemcpyDeviceToHost, ctx->side));
    PCV_HIP_CHECK(ctx, hipEventRecord(ctx->spec_ev, ctx->side));

    // ---- meanwhile: the host's view of the tree ----
    PCV_HIP_CHECK(ctx, hipEventSynchronize(ctx->spec_ev));
    std::memcpy(info, hs, sizeof(info));
    if (info[1] & 2u) return PCV_OK;  // table capacity: let the exact pipeline report it
    if (info[1] & 1u) {               // deeper than the sample keys
      if (sample_levels < full_levels) {
        sample_levels = full_levels;
        ctx->stage_begin(PCV_STAGE_CHAIN_KEYS);
        continue;
      }
      return PCV_OK;  // deeper than one key word: the exact pipeline (deep path) takes it
    }
    const uint32_t tn = info[0];
    if (tn <= first) {
      if (!pcv_spec_tree_from_walk((const uint32_t*)(hs + h_walk), (const uint32_t*)(hs + h_parent), hs + h_level, tn, &tree))
        return ctx->fail(PCV_E_HIP, "single-chain build: inconsistent predicted tree");
    } else {  // a big tree: fetch all of it (queued behind the chain pass; 500 M+ points)
      std::vector<uint32_t> w(tn), pr(tn);
      std::vector<uint8_t> lvl(tn);
      PCV_HIP_CHECK(ctx, hipMemcpyAsync(w.data(), d_walk, (size_t)tn * 4, hipMemcpyDeviceToHost, st));
      PCV_HIP_CHECK(ctx, hipMemcpyAsync(pr.data(), d_sparent, (size_t)tn * 4, hipMemcpyDeviceToHost, st));
      PCV_HIP_CHECK(ctx, hipMemcpyAsync(lvl.data(), d_slevel, (size_t)tn, hipMemcpyDeviceToHost, st));
      PCV_HIP_CHECK(ctx, hipStreamSynchronize(st));
      if (!pcv_spec_tree_from_walk(w.data(), pr.data(), lvl.data(), tn, &tree))
        return ctx->fail(PCV_E_HIP, "single-chain build: inconsistent predicted tree");
    }
    break;
  }
  sp.nlevels = sample_levels;

  // ---- exact counts -> rank map (device) and true tree (host) ----
  ctx->stage_begin(PCV_STAGE_NODE_SPLIT);
  // 12-byte records and a predicted tree whose counters fit one LDS histogram and whose map fits beside the sort's staging
  // (up to 16 384 nodes: clouds of up to ~250 M points at the default capacity):
  // the count runs over the record sort's own workgroups and keeps every workgroup's histogram, from which the sort's first
  // pass derives its digit histogram (the keys are then read by its downsweep only, which applies the map itself).
  // PCV_SORT_ROWS=0 (libpcv_hip_exp.so): the first pass counts (and maps) the keys in a pass of its own.
  static const bool rows_on = [] {
    const char* e = pcv_experiment("PCV_SORT_ROWS");
    return !e || atoi(e) != 0;
  }();
  bs->spec_rows = nullptr;
  if (rows_on && compact && tree.num_leaves <= pcv_rank_hist_max_bins()) {
    int sgroups;
    uint64_t schunk;
    pcv_sort_rec12_geometry(n, &sgroups, &schunk);
    uint32_t* rows = nullptr;
    // the rows are an optimisation (up to ~1 GB beside the records for the biggest trees): a failed allocation falls
    // back to the plain count, whose sort takes a counting pass instead
    if (sc.get(&rows, (size_t)sgroups * tree.num_leaves) == PCV_OK) {
      pcv_launch_rank_hist_rows(ctx, rank, n, tree.num_leaves, d_counts, 8, sgroups, schunk, rows);
      bs->spec_rows = rows;
    }
  }
  if (!bs->spec_rows) pcv_launch_rank_hist(ctx, rank, n, tree.num_leaves, d_counts, compact ? 8 : 0);
  PCV_HIP_CHECK(ctx, hipGetLastError());
  if ((rc = ctx->pinned_spec_reserve((size_t)tree.num_leaves * 8 + 512 + kPcvPoolRegions * 4))) return rc;
  uint8_t* hp = (uint8_t*)ctx->pinned_spec;
  // the number of `wide` pool entries the chain pass handed out per region (the record epilogue of chain_pass_kernel), then the exact counts: one
  // block on the device, one copy (every small operation on `stream` costs a hand-over of ~10 us between two kernels)
  uint32_t* h_pool = (uint32_t*)hp;
  uint32_t* h_counts = h_pool + kPcvPoolRegions;
  const size_t map_off = (((size_t)kPcvPoolRegions + tree.num_leaves) * 4 + 255) & ~(size_t)255;
  // (the copy travels on the side stream, behind an event of `stream`: a copy command between the count and the resolve kernel costs
  // `stream` two hand-overs of ~10 us; nothing queued on `stream` later writes these counters)
  if (ctx->side_begin() != PCV_OK) return ctx->fail(PCV_E_HIP, "single-chain build: side stream");
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(h_pool, d_pool_ctr, ((size_t)kPcvPoolRegions + tree.num_leaves) * 4, hipMemcpyDeviceToHost, ctx->side));
  PCV_HIP_CHECK(ctx, hipEventRecord(ctx->spec_ev, ctx->side));  // the counts are on their way to the host
  // The map the record sort needs is computed on the device (spec_resolve_kernel), and the sort is queued behind it right
  // away: the counts' trip to the host, the host's own resolve and the table building all happen beside the sort instead
  // of in front of it. The sort's digit widths come from the number of PREDICTED leaves (an upper bound of the true
  // leaves, known since the host mirrored T''). PCV_DEVICE_RESOLVE=0 (libpcv_hip_exp.so): the map comes from the host.
  static const bool device_resolve = [] {
    const char* e = pcv_experiment("PCV_DEVICE_RESOLVE");
    return !e || atoi(e) != 0;
  }();
  bs->spec_wide = wide;
  bs->wide_levels = wide_levels;
  bs->spec_map_dev = d_map;
  bs->spec_map_entries = tree.num_leaves;
  bs->spec = true;
  bs->spec_payload = payload;
  bs->fix_ranges.clear();
  auto give_up = [&]() {  // nothing of this attempt is kept; queued work on the scratch buffers drains harmlessly
    // (the pool is stream-ordered: a freed block is only handed to work queued later on the same stream). The record sort
    // that was queued ahead of the verdict leaves its second payload buffer and the rank-count rows behind: without this
    // the fallback — which allocates its own record buffers — would be the build's memory peak
    if (bs->sort_queued && bs->pay_b && bs->pay_b != (void*)payload) {
      sc.detach(bs->pay_b);
      ctx->dev_free(bs->pay_b);
    }
    if (bs->spec_rows) {
      sc.detach((void*)bs->spec_rows);
      ctx->dev_free((void*)bs->spec_rows);
    }
    bs->pay_a = bs->pay_b = nullptr;
    bs->spec = false;
    bs->spec_payload = nullptr;
    bs->spec_wide = nullptr;
    bs->spec_map_dev = nullptr;
    bs->spec_rows = nullptr;
    bs->sort_queued = false;
    // a held-back second pass is simply never queued — but its layout kernels may still be running on the side stream, writing
    // into the sort scratch that goes back to the (main-stream-ordered) pool below: join them first
    if (bs->sort_second.pending && bs->sort_second.join_side) (void)ctx->side_end();
    bs->sort_second = PcvSortSecond();
    bs->resolve_on_device = false;
    sc.detach(payload);
    ctx->dev_free(payload);
    if (wide) {
      sc.detach(wide);
      ctx->dev_free(wide);
    }
  };
  if (device_resolve) {
    uint32_t *d_nst, *d_base, *d_out;
    const uint32_t tn = (uint32_t)tree.prefix.size();
    if ((rc = sc.get(&d_nst, tn)) || (rc = sc.get(&d_base, tn))) return rc;
    // the kernel stores its verdict straight into the pinned mailbox (host memory the device can write): no copy between
    // the resolve kernel and the sort
    d_out = (uint32_t*)(ctx->mailbox_dev + kMailboxResolve);
    // the sentinel goes in BEFORE the launch: nothing queued on the stream writes this slot until the kernel does, and a
    // store after the launch could overwrite a verdict the kernel had already delivered
    ctx->mailbox[kMailboxResolve] = ~0ull;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    pcv_launch_spec_resolve(ctx, lv, params->resolution, sp.cap, sp.force_mask, d_walk, d_slevel, tn, d_counts, d_nst, d_base, d_map, d_out);
    PCV_HIP_CHECK(ctx, hipGetLastError());  // before the sort is queued behind it
    // {true leaves, too shallow} as the device sees them: read by pcv_build_finish (which synchronises anyway) and held
    // against the host's resolve — the device map drives the sort, the host's tree the tables
    bs->resolve_on_device = true;
    ctx->stage_end(PCV_STAGE_NODE_SPLIT);
    const uint32_t predicted_leaves = (uint32_t)std::count(tree.inner.begin(), tree.inner.end(), (uint8_t)0);
    if ((rc = queue_record_sort(ctx, bs, t, nullptr, predicted_leaves, false))) return rc;
    host_lap("record sort queued");
  }
  PCV_HIP_CHECK(ctx, hipEventSynchronize(ctx->spec_ev));
  host_lap("", true);
  const PcvSpecStatus resolved = pcv_spec_resolve(sp, tree, h_counts, tt);
  host_lap("resolve");
  if (resolved != PCV_SPEC_OK) {
    give_up();
    return PCV_OK;  // *used stays false
  }
  // the rare replay takes its pool entries from the TOPS of the pool regions (spec_replay_kernel: slot j of the flattened
  // replay list -> region j % regions): they must not reach down to what the chain pass filled from the bottoms. A point
  // has at most one live entry, but a replayed point may have used one in the chain pass already — on an adversarial
  // cloud a region can run out: the exact pipeline takes the build
  if (wide) {
    uint64_t replay_slots = 0, used = 0, most = 0;
    for (uint32_t k : tt->fix_nodes) replay_slots += tt->hi[k] - tt->lo[k];
    for (uint32_t r = 0; r < kPcvPoolRegions; ++r) {
      used += h_pool[r];
      most = std::max<uint64_t>(most, h_pool[r]);
    }
    if (most + (replay_slots + kPcvPoolRegions - 1) / kPcvPoolRegions > pool_cap) {
      give_up();
      return PCV_OK;
    }
    t->wide_pool_entries = used;
  }
  bs->resolve_host_leaves = tt->num_leaves;
  bs->resolve_check_map = (params->flags & PCV_BUILD_CHECK_RESOLVE) != 0;
  if (bs->resolve_check_map) {
    bs->resolve_host_map = tt->spec_map;
    bs->resolve_host_inner = tree.inner;
  }
  if (tt->prefix.size() > (size_t)nt.capacity) return ctx->fail(PCV_E_OOM, "node table capacity exceeded");
  bs->cont_nodes = tt->cont_nodes;
  bs->cont_from = tt->cont_from;
  // leaves whose points still have to replay the chain: contiguous once the records are sorted ([lo, hi) of the leaf)
  for (uint32_t k : tt->fix_nodes) bs->fix_ranges.push_back({tt->lo[k], tt->hi[k] - tt->lo[k], (uint32_t)tt->level[k]});
  if (!device_resolve) {
    // the host's map goes up and is applied by the first upsweep of the record sort; no synchronisation: the upload reads
    // ctx->pinned_spec, the caller stages the node table in ctx->pinned
    std::memcpy(hp + map_off, tt->spec_map.data(), (size_t)tree.num_leaves * 4);
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(d_map, hp + map_off, (size_t)tree.num_leaves * 4, hipMemcpyHostToDevice, st));
    ctx->stage_end(PCV_STAGE_NODE_SPLIT);
    host_lap("map upload, fix ranges");
    if ((rc = queue_record_sort(ctx, bs, t, nullptr, tt->num_leaves, false))) return rc;
    host_lap("record sort queued");
  }
  // replayed leaves rewrite their SORTED records: with the sort's second pass held back (PcvSortSecond) the replay is queued
  // behind that pass, in pcv_build_finish
  if (!bs->sort_second.pending && (rc = queue_replay(ctx, bs))) return rc;
  ctx->stage_begin(PCV_STAGE_TABLE);
  t->spec_stats[0] = tree.prefix.size();
  t->spec_stats[1] = (uint64_t)std::count(tree.inner.begin(), tree.inner.end(), (uint8_t)0);
  t->spec_stats[2] = tt->kept_points;
  t->spec_stats[3] = tt->fix_points;
  t->spec_continued = tt->cont_points;
  *used = true;
  return PCV_OK;
}

static int build_begin_impl(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points,
                            const pcv_routed_points* routed, pcv_octree** out, const PcvTrueTree* given_tree) {
  int rc;
  if (!(params->resolution > 0.0) || !std::isfinite(params->resolution)) return ctx->fail(PCV_E_INVALID, "resolution must be a positive finite number");
  const uint32_t max_points = params->max_points_per_node ? params->max_points_per_node : PCV_DEFAULT_MAX_POINTS_PER_NODE;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint64_t n = routed ? routed->n : points->n;

  PcvBuild* bs = new PcvBuild(ctx);
  pcv_octree* t = new pcv_octree();
  t->pending = bs;  // owned by the tree from here on
  t->ctx = ctx;
  PcvScratch& sc = bs->sc;
  DevPoints& d = bs->d;
  bs->n = n;
  t->resolution = params->resolution;
  t->has_intensity = (routed ? routed->intensity : points->intensity) != nullptr;
  struct Guard {
    pcv_octree* t;
    ~Guard() {
      if (t) pcv_octree_free(t);
    }
  } guard{t};
  PCV_HIP_CHECK(ctx, hipEventRecord(ctx->ev[0], st));
  for (bool& on : ctx->stage_on) on = false;
  for (bool& open : ctx->stage_open) open = false;
  ctx->stage_times = bs->stage_times = (params->flags & PCV_BUILD_STAGE_TIMES) != 0;
  ctx->stage_begin(PCV_STAGE_AABB);
  if (routed) {  // device-resident by contract
    d.n = n;
    d.routed.oct = reinterpret_cast<const uint8_t*>(routed->oct_rgb);  // byte 0 of every packed word
    d.routed.oct_stride = 4;
    d.routed.cx = routed->cx;
    d.routed.cy = routed->cy;
    d.routed.cz = routed->cz;
    d.color = reinterpret_cast<const uint8_t*>(routed->oct_rgb) + 1;  // r, g, b follow the digit
    d.color_stride = 4;
    d.intensity = routed->intensity;
  } else if ((rc = stage_points(ctx, sc, points, true, &d))) {
    return rc;
  }

  double bmin[3], bmax[3];
  if ((params->flags & PCV_BUILD_COMPUTE_BBOX) && n > 0) {
    // K1 is queued; the build's three big scratch blocks are taken from the pool while it runs (the host has nothing else to do
    // for 0.4 ms at 100 M points, and these calls are the slow ones of what follows the box)
    if ((rc = device_aabb_launch(ctx, sc, d))) return rc;
    if ((rc = sc.get(&bs->keys_a, n)) || (rc = sc.get(&bs->keys_b, n))) return rc;
    if ((rc = ctx->dev_alloc(&bs->sort_scratch, pcv_sort_scratch_bytes(n)))) return rc;
    sc.ptrs.push_back(bs->sort_scratch);
    if ((rc = device_aabb_wait(ctx, bmin, bmax))) return rc;
  } else if (params->flags & PCV_BUILD_COMPUTE_BBOX) {
    if ((rc = device_aabb(ctx, sc, d, bmin, bmax))) return rc;
  } else {
    for (int a = 0; a < 3; ++a) {
      bmin[a] = params->bbox_min[a];
      bmax[a] = params->bbox_max[a];
    }
  }
  for (int a = 0; a < 3; ++a) {
    t->bbox_min[a] = bmin[a];
    t->bbox_max[a] = bmax[a];
  }
  ctx->stage_end(PCV_STAGE_AABB);
  if (n == 0) {  // generation.rs:325-330: no leaves, no finished nodes, meta without nodes
    delete t->pending;
    t->pending = nullptr;
    *out = t;
    guard.t = nullptr;
    return PCV_OK;
  }

  PcvLevels& lv = bs->lv;
  int max_level = 0;
  pcv_make_levels(bmin, bmax, params->resolution, 64, &lv, &max_level, nullptr, nullptr);
  if (routed && (lv.nlevels < 1 || lv.enc[1] != PCV_ENC_FLOAT32))
    return ctx->fail(PCV_E_INVALID, "routed points carry Float32 level-1 codes, but level 1 of this cube is not Float32-encoded");

  ctx->stage_begin(PCV_STAGE_CHAIN_KEYS);
  // ---- K2 keys, K3 sort, K4 node split — with depth speculation ----
  // The keys only have to cover the levels the tree really uses. A strided sample (2^18 points) gets full-depth
  // keys, is sorted, and depth_probe measures the deepest prefix still shared by sample keys `gap` apart
  // (gap = 0.6 x the sample-scaled node capacity, i.e. biased towards deeper). The main pass then computes and
  // sorts only that many levels (+1), with 32-bit keys when 10 levels suffice. K4 verifies: if any node at the last
  // key level would still have to be split, everything is redone at full depth — speculation can cost time, never
  // correctness.
  uint64_t*& keys_a = bs->keys_a;
  uint64_t*& keys_b = bs->keys_b;
  void*& sort_scratch = bs->sort_scratch;
  if (!keys_a) {  // (taken while K1 ran when the build computes its own box)
    if ((rc = sc.get(&keys_a, n)) || (rc = sc.get(&keys_b, n))) return rc;
    if ((rc = ctx->dev_alloc(&sort_scratch, pcv_sort_scratch_bytes(n)))) return rc;
    sc.ptrs.push_back(sort_scratch);
  }
  const int full_levels = lv.nlevels;
  int spec_levels = full_levels;

  // device node table (the sample tree of the single-chain build, the tree itself in the exact build)
  PcvNodeTableDev nt;
  {
    const int deepest = max_level < PCV_MAX_LEVELS ? max_level : PCV_MAX_LEVELS;  // incl. the deep retry
    uint64_t cap64 = 8ull * (uint64_t)(deepest + 1) * (n / max_points + 1) + 64;
    if (cap64 > (1ull << 26)) cap64 = 1ull << 26;
    const uint32_t cap = (uint32_t)cap64;
    nt.capacity = cap;
    nt.prefix_lo = nullptr;
    if ((rc = sc.get(&nt.prefix, cap)) || (rc = sc.get(&nt.lo, cap)) || (rc = sc.get(&nt.hi, cap)) ||
        (rc = sc.get(&nt.parent, cap)) || (rc = sc.get(&nt.first_child, cap)) || (rc = sc.get(&nt.level, cap)) ||
        (rc = sc.get(&nt.child_mask, cap)) || (rc = sc.get(&nt.open, cap)) ||
        (rc = sc.get(&nt.counters, 64)))
      return rc;
    // (the sample tree of the single-chain build splits at a LOWERED threshold: twice the open nodes the capacity allows)
    nt.max_open = (uint32_t)std::min<uint64_t>(2 * (n / max_points + 1) + 64, 1u << 22);
    if ((rc = sc.get(&nt.bounds, std::max((size_t)cap * 9, (size_t)nt.max_open * 85)))) return rc;
  }
  uint32_t counters[64];
  bool keys32 = false;
  int attempts = 0;

  // ---- single-chain build (pcv_spec.h): ONE chain pass, topology from a sample + exact per-leaf counts ----
  bool spec_used = false;
  PcvTrueTree true_tree;
  if (given_tree) {  // pcv_gather_encode: the topology is an input; K5 / K6 run on it as they do after K4
    true_tree = *given_tree;
    spec_used = true;
    counters[0] = (uint32_t)true_tree.prefix.size();
    ctx->stage_begin(PCV_STAGE_TABLE);
  } else {
    bool wide_level = false;  // a Float64-encoded level needs the high code words: left to the exact pipeline
    for (int k = 0; k <= full_levels; ++k) wide_level = wide_level || lv.enc[k] == PCV_ENC_FLOAT64;
    const bool forced = (params->flags & PCV_BUILD_FORCE_SINGLE_CHAIN) != 0;
    const bool want = !(params->flags & (PCV_BUILD_NO_SPECULATION | PCV_BUILD_NO_SINGLE_CHAIN)) && (forced || n >= (1ull << 22));
    if (want && !wide_level && full_levels >= 1) {
      if ((rc = single_chain_topology(ctx, bs, t, params, max_points, full_levels, nt, &true_tree, &spec_used))) return rc;
      if (spec_used) counters[0] = (uint32_t)true_tree.prefix.size();
      else ++attempts;  // the prediction was too shallow somewhere: the exact pipeline redoes the build
    }
  }

  if (!spec_used && n >= (1ull << 22) && full_levels > 4 && !(params->flags & PCV_BUILD_NO_SPECULATION)) {
    const uint32_t ns = 1u << 18;
    const uint64_t stride = n / ns;
    uint32_t* d_max;
    if ((rc = sc.get(&d_max, 64))) return rc;
    PCV_HIP_CHECK(ctx, hipMemsetAsync(d_max, 0, 4, st));
    pcv_launch_chain_keys(ctx, lv, ns, stride, d.x, d.y, d.z, keys_a, false, d.routed);
    bool s_in_a = true;
    // the probe only has to tell depths up to kProbeLevels apart (deeper -> no speculation): sort those digits only
    const int kProbeLevels = 14;
    const int probe_levels = full_levels < kProbeLevels ? full_levels : kProbeLevels;
    if ((rc = pcv_radix_sort_u64(ctx, keys_a, keys_b, ns, 3 * (PCV_MAX_KEY_LEVELS - probe_levels), 3 * PCV_MAX_KEY_LEVELS,
                                 nullptr, sort_scratch, &s_in_a)))
      return rc;
    double gapd = 0.6 * (double)max_points * (double)ns / (double)n;
    uint32_t gap = gapd < 1.0 ? 1u : (uint32_t)gapd;
    pcv_launch_depth_probe(ctx, s_in_a ? keys_a : keys_b, ns, gap, d_max);
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->mailbox, d_max, 4, hipMemcpyDeviceToHost, st));
    PCV_HIP_CHECK(ctx, hipStreamSynchronize(st));
    const uint32_t shared = *(const uint32_t*)ctx->mailbox;
    // a level-`shared` node is (probably) split -> nodes of level shared + 1 exist -> that many digits are needed;
    // a prefix shared down to the last sorted level says nothing about the levels below it
    int want = (int)shared >= probe_levels ? full_levels : (int)shared + 1;
    if (want < 3) want = 3;
    if (want < full_levels) spec_levels = want;
  }

  for (; !spec_used;) {
    ++attempts;
    lv.nlevels = spec_levels;
    keys32 = spec_levels <= 10;
    pcv_launch_chain_keys(ctx, lv, n, 1, d.x, d.y, d.z, keys_a, keys32, d.routed);
    ctx->stage_end(PCV_STAGE_CHAIN_KEYS);
    ctx->stage_begin(PCV_STAGE_SORT_KEYS);
    bool in_a = true;
    if (keys32)
      rc = pcv_radix_sort_u32(ctx, (uint32_t*)keys_a, (uint32_t*)keys_b, n, 3 * (10 - spec_levels), 30, nullptr,
                              sort_scratch, &in_a);
    else
      rc = pcv_radix_sort_u64(ctx, keys_a, keys_b, n, 3 * (PCV_MAX_KEY_LEVELS - spec_levels), 3 * PCV_MAX_KEY_LEVELS,
                              nullptr, sort_scratch, &in_a);
    if (rc) return rc;
    const void* sorted_keys = in_a ? (const void*)keys_a : (const void*)keys_b;
    ctx->stage_end(PCV_STAGE_SORT_KEYS);
    ctx->stage_begin(PCV_STAGE_NODE_SPLIT);

    // K4: every open node holds > max_points points and open nodes of one level are disjoint
    pcv_launch_node_split(ctx, nt, sorted_keys, keys32, (uint32_t)n, lv, params->resolution, max_points,
                          (params->flags >> 8) & 0xffu);
    ctx->stage_end(PCV_STAGE_NODE_SPLIT);
    ctx->stage_begin(PCV_STAGE_TABLE);

    // ---- node table to host ----
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->mailbox, nt.counters, sizeof(counters), hipMemcpyDeviceToHost, st));
    PCV_HIP_CHECK(ctx, hipStreamSynchronize(st));
    std::memcpy(counters, ctx->mailbox, sizeof(counters));
    if (counters[1] & 2u) return ctx->fail(PCV_E_OOM, "node table capacity exceeded");
    if (counters[1] & 1u) {
      if (spec_levels < full_levels) {  // speculation too shallow: redo at full depth
        spec_levels = full_levels;
        continue;
      }
      if (max_level > PCV_MAX_KEY_LEVELS) {
        // ---- deep tree: a second key word for levels 22..40 (heavy duplicates in a cube with edge/resolution > 2^21).
        // Rare path, built from existing pieces: digits as four 32-bit words, four stable LSD sorts (one per word,
        // least significant first, the other three travel as payload planes), then the split with both words.
        ++attempts;
        const int deep_levels = max_level < PCV_MAX_LEVELS ? max_level : PCV_MAX_LEVELS;
        bs->deep = true;
        lv.nlevels = deep_levels;
        uint32_t* W[4][2];
        for (int j = 0; j < 4; ++j)
          for (int sd = 0; sd < 2; ++sd)
            if ((rc = sc.get(&W[j][sd], n))) return rc;
        uint32_t* first[4] = {W[0][0], W[1][0], W[2][0], W[3][0]};
        pcv_launch_chain_keys_deep(ctx, lv, n, d.x, d.y, d.z, d.routed, first);
        int side = 0;
        for (int j = 3; j >= 0; --j) {
          PcvSortPayload pl;
          pl.nwords = 3;
          int w = 0;
          for (int o = 0; o < 4; ++o) {
            if (o == j) continue;
            pl.in[w] = W[o][side];
            pl.out[w] = W[o][1 - side];
            ++w;
          }
          bool in_a = true;
          if ((rc = pcv_radix_sort_u32(ctx, W[j][side], W[j][1 - side], n, 0, 32, &pl, sort_scratch, &in_a))) return rc;
          if (!in_a) side = 1 - side;
        }
        const uint32_t* sorted_words[4] = {W[0][side], W[1][side], W[2][side], W[3][side]};
        pcv_launch_combine_words(ctx, n, sorted_words, keys_a, keys_b);  // keys_a = first word, keys_b = second word
        if ((rc = sc.get(&nt.prefix_lo, nt.capacity))) return rc;
        bs->d_prefix_lo = nt.prefix_lo;
        pcv_launch_node_split(ctx, nt, keys_a, false, (uint32_t)n, lv, params->resolution, max_points,
                              (params->flags >> 8) & 0xffu, keys_b);
        ctx->stage_end(PCV_STAGE_NODE_SPLIT);
        ctx->stage_begin(PCV_STAGE_TABLE);
        PCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->mailbox, nt.counters, sizeof(counters), hipMemcpyDeviceToHost, st));
        PCV_HIP_CHECK(ctx, hipStreamSynchronize(st));
        std::memcpy(counters, ctx->mailbox, sizeof(counters));
        if (counters[1] & 2u) return ctx->fail(PCV_E_OOM, "node table capacity exceeded");
        if (!(counters[1] & 1u)) {
          spec_levels = deep_levels;
          break;
        }
      }
      return ctx->fail(PCV_E_DEPTH, "a node at level " + std::to_string(lv.nlevels) +
                                        " still holds more than max_points_per_node points and is larger than the "
                                        "resolution; the reference's NodeId cannot name deeper nodes either");
    }
    break;
  }
  t->key_levels = spec_levels;
  t->key_attempts = attempts;
  bs->levels = bs->deep ? lv.nlevels : full_levels;
  lv.nlevels = bs->levels;  // K5/K6 index the level tables by node level; the walk stops at leaves anyway
  const uint32_t M = counters[0];
  bs->M = M;
  // pinned staging: prefix(8) lo hi parent first_child (4 each) level mask open (1 each)
  const size_t lo_off = (((size_t)M * (8 + 4 * 4 + 3) + 64) + 7) & ~(size_t)7;  // deep trees: second prefix word
  const size_t host_bytes = lo_off + (bs->deep ? (size_t)M * 8 : 0);
  bs->host_bytes = host_bytes;
  if ((rc = ctx->pinned_reserve(host_bytes * 4 + (size_t)M * 72 + (size_t)M * 2 * sizeof(PcvNodeRec) + 1024 +
                                (2 * (bs->n / kPcvSettleTile) + bs->n / (8 * kPcvClimbTile) + 3 * (size_t)M + 8) * sizeof(PcvSettleItem) +
                                (size_t)M * pcv_cont_range_bytes() + (size_t)M + 64)))
    return rc;
  uint8_t* hp = (uint8_t*)ctx->pinned;
  uint64_t* h_prefix = (uint64_t*)hp;
  uint32_t* h_lo = (uint32_t*)(h_prefix + M);
  uint32_t* h_hi = h_lo + M;
  uint32_t* h_first = h_hi + M;
  uint8_t* h_level = (uint8_t*)(h_first + M);
  uint8_t* h_mask = h_level + M;
  uint8_t* h_open = h_mask + M;
  if (spec_used) {  // the true tree was derived on the host from the exact per-leaf counts: same layout, no download
    std::memcpy(h_prefix, true_tree.prefix.data(), (size_t)M * 8);
    std::memcpy(h_lo, true_tree.lo.data(), (size_t)M * 4);
    std::memcpy(h_hi, true_tree.hi.data(), (size_t)M * 4);
    std::memcpy(h_first, true_tree.first_child.data(), (size_t)M * 4);
    std::memcpy(h_level, true_tree.level.data(), (size_t)M);
    std::memcpy(h_mask, true_tree.child_mask.data(), (size_t)M);
    std::memcpy(h_open, true_tree.open.data(), (size_t)M);
  } else {
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(h_prefix, nt.prefix, (size_t)M * 8, hipMemcpyDeviceToHost, st));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(h_lo, nt.lo, (size_t)M * 4, hipMemcpyDeviceToHost, st));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(h_hi, nt.hi, (size_t)M * 4, hipMemcpyDeviceToHost, st));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(h_first, nt.first_child, (size_t)M * 4, hipMemcpyDeviceToHost, st));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(h_level, nt.level, (size_t)M, hipMemcpyDeviceToHost, st));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(h_mask, nt.child_mask, (size_t)M, hipMemcpyDeviceToHost, st));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(h_open, nt.open, (size_t)M, hipMemcpyDeviceToHost, st));
    if (bs->deep)
      PCV_HIP_CHECK(ctx, hipMemcpyAsync(hp + lo_off, nt.prefix_lo, (size_t)M * 8, hipMemcpyDeviceToHost, st));
    PCV_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }

  // upload area (pinned, after the download area)
  uint8_t* up = hp + ((host_bytes + 255) & ~(size_t)255);
  uint64_t* u_walk = (uint64_t*)up;                 // M
  uint64_t* u_xyz_off = u_walk + M;                  // M
  uint64_t* u_point_off = u_xyz_off + M;             // M
  double* u_node_min = (double*)(u_point_off + M);   // 3M
  uint32_t* u_parent = (uint32_t*)(u_node_min + 3 * (size_t)M);  // M
  uint32_t* u_child_off = u_parent + M;              // M
  uint32_t* u_leaf_lo = u_child_off + M;             // <= M
  uint32_t* u_leaf_node = u_leaf_lo + M;             // <= M
  uint8_t* u_level = (uint8_t*)(u_leaf_node + M);    // M
  (void)u_walk; (void)u_leaf_lo; (void)u_level;

  std::vector<uint64_t>& pre = bs->pre;
  pre.assign(M, 0);
  // bottom-up stream lengths: |pre(inner)| = sum ceil(|pre(child)| / 8)
  for (uint32_t i = M; i-- > 0;) {
    if (!h_open[i]) {
      pre[i] = (uint64_t)h_hi[i] - h_lo[i];
    } else {
      uint64_t acc = 0;
      uint32_t c = h_first[i];
      for (int dgt = 0; dgt < 8; ++dgt)
        if ((h_mask[i] >> dgt) & 1) {
          u_child_off[c] = (uint32_t)acc;
          u_parent[c] = i;
          acc += ceil8(pre[c]);
          ++c;
        }
      pre[i] = acc;
    }
  }
  u_parent[0] = 0xffffffffu;
  u_child_off[0] = 0;
  *out = t;
  guard.t = nullptr;
  return PCV_OK;
}

extern "C" int pcv_build_top_streams(const pcv_octree* t, pcv_top_streams* out) {
  if (!t || !out) return PCV_E_INVALID;
  std::memset(out, 0, sizeof(*out));
  const PcvBuild* bs = t->pending;
  if (!bs) return t->nodes.empty() ? PCV_OK : t->ctx->fail(PCV_E_INVALID, "pcv_build_top_streams needs a tree between pcv_build_begin and pcv_build_finish");
  pcv_ctx* ctx = t->ctx;
  const uint32_t M = bs->M;
  const uint8_t* hp = (const uint8_t*)ctx->pinned;  // the node table staged by pcv_build_begin
  const uint32_t* h_first = (const uint32_t*)(hp + (size_t)M * 16);
  const uint8_t* h_mask = hp + (size_t)M * 21;
  const uint8_t* h_open = hp + (size_t)M * 22;
  uint32_t c1 = h_first[0];
  for (int c = 0; c < 8; ++c) {
    if (!((h_mask[0] >> c) & 1)) continue;
    const uint32_t i = c1++;
    out->l1[c] = bs->pre[i];
    if (!h_open[i]) continue;
    out->l1_split_mask |= 1u << c;
    uint32_t c2 = h_first[i];
    for (int dg = 0; dg < 8; ++dg)
      if ((h_mask[i] >> dg) & 1) out->l2[c * 8 + dg] = bs->pre[c2++];
  }
  return PCV_OK;
}

extern "C" int pcv_build_finish(pcv_octree* t, const pcv_top_layout* top) {
  if (!t) return PCV_E_INVALID;
  PcvBuild* bs = t->pending;
  if (!bs) return t->nodes.empty() && t->num_points == 0 ? PCV_OK : t->ctx->fail(PCV_E_INVALID, "pcv_build_finish: nothing pending");
  pcv_ctx* ctx = t->ctx;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rc;
  struct Done {  // the build state is released on every exit path; on failure the tree stays valid but empty
    pcv_octree* t;
    ~Done() {
      delete t->pending;
      t->pending = nullptr;
    }
  } done{t};
  PcvScratch& sc = bs->sc;
  PcvLevels& lv = bs->lv;
  const uint64_t n = bs->n;
  const uint32_t M = bs->M;
  const size_t host_bytes = bs->host_bytes;
  ctx->stage_times = bs->stage_times;  // the flag of this build, whatever the context has begun since

  uint64_t* keys_a = bs->keys_a;
  uint64_t* keys_b = bs->keys_b;
  std::vector<uint64_t>& pre = bs->pre;
  double bmin[3] = {t->bbox_min[0], t->bbox_min[1], t->bbox_min[2]};
  uint8_t* hp = (uint8_t*)ctx->pinned;
  uint64_t* h_prefix = (uint64_t*)hp;
  uint32_t* h_lo = (uint32_t*)(h_prefix + M);
  uint32_t* h_hi = h_lo + M;
  uint32_t* h_first = h_hi + M;
  uint8_t* h_level = (uint8_t*)(h_first + M);
  uint8_t* h_mask = h_level + M;
  uint8_t* h_open = h_mask + M;
  // upload area (pinned, after the download area)
  uint8_t* up = hp + ((host_bytes + 255) & ~(size_t)255);
  uint64_t* u_walk = (uint64_t*)up;                 // M
  uint64_t* u_xyz_off = u_walk + M;                  // M
  uint64_t* u_point_off = u_xyz_off + M;             // M
  double* u_node_min = (double*)(u_point_off + M);   // 3M
  uint32_t* u_parent = (uint32_t*)(u_node_min + 3 * (size_t)M);  // M
  uint32_t* u_child_off = u_parent + M;              // M
  uint32_t* u_leaf_lo = u_child_off + M;             // <= M
  uint32_t* u_leaf_node = u_leaf_lo + M;             // <= M
  uint8_t* u_level = (uint8_t*)(u_leaf_node + M);    // M
  const size_t up_bytes = (size_t)M * (8 * 3 + 24 + 4 * 4 + 1);
  const uint64_t* h_prefix_lo = (const uint64_t*)(hp + ((((size_t)M * (8 + 4 * 4 + 3) + 64) + 7) & ~(size_t)7));  // deep only

  uint32_t top_nodes = 0;  // nodes of level <= 1 hold GLOBAL streams when a layout is given (multi-GPU build)
  if (top) {
    pre[0] = top->root_points;
    uint32_t c1 = h_first[0];
    top_nodes = 1;
    for (int c = 0; c < 8; ++c) {
      if (!((h_mask[0] >> c) & 1)) continue;
      const uint32_t i = c1++;
      ++top_nodes;
      pre[i] = top->l1_stream[c];
      u_child_off[i] = top->l1_offset[c];
      if (!h_open[i]) continue;
      uint32_t c2 = h_first[i];
      for (int dg = 0; dg < 8; ++dg)
        if ((h_mask[i] >> dg) & 1) u_child_off[c2++] = top->l2_offset[c * 8 + dg];
    }
  }
  // leaves in key order == order of their sorted ranges: depth-first, children in digit order (no sort needed)
  std::vector<uint32_t> leaves;
  leaves.reserve(M);
  {
    std::vector<uint32_t> stack;
    stack.reserve(8 * (PCV_MAX_KEY_LEVELS + 1));
    stack.push_back(0);
    while (!stack.empty()) {
      const uint32_t i = stack.back();
      stack.pop_back();
      if (!h_open[i]) {
        leaves.push_back(i);
        continue;
      }
      const uint32_t nchild = (uint32_t)__builtin_popcount(h_mask[i]);
      for (uint32_t c = nchild; c-- > 0;) stack.push_back(h_first[i] + c);  // reversed: digit 0 is popped first
    }
  }
  const uint32_t num_leaves = (uint32_t)leaves.size();
  std::vector<uint32_t> rank_of(M, 0);
  bool wide = false;
  for (uint32_t r = 0; r < num_leaves; ++r) {
    uint32_t i = leaves[r];
    rank_of[i] = r;
    u_leaf_lo[r] = h_lo[i];
    u_leaf_node[r] = i;
    if (lv.enc[h_level[i]] == PCV_ENC_FLOAT64) wide = true;
  }
  // The single-chain build queued its record sort as soon as the rank map was on the device (queue_record_sort): the
  // host work from here to K6 — node tables and their upload — hides behind the two sort passes.
  const int w_int = t->has_intensity ? 0 : -1;
  const int w_hi = wide ? (t->has_intensity ? 1 : 0) : -1;
  PcvWalkTables wt;
  if (bs->spec && !bs->sort_queued) {
    ctx->stage_end(PCV_STAGE_TABLE);
    if ((rc = queue_record_sort(ctx, bs, t, nullptr, num_leaves, wide))) return rc;
  }
  t->nodes.resize(M);
  uint64_t xyz_off = 0, point_off = 0;
  for (uint32_t i = 0; i < M; ++i) {
    const int level = h_level[i];
    u_level[i] = (uint8_t)level;
    u_walk[i] = h_open[i] ? ((uint64_t)h_first[i] | ((uint64_t)h_mask[i] << 32) | ((uint64_t)level << 48))
                          : ((uint64_t)rank_of[i] | (1ull << 40) | ((uint64_t)level << 48));
    // NodeId::find_bounding_cube recurrence (node.rs:157-172): parents precede children in the table
    if (i == 0) {
      for (int a = 0; a < 3; ++a) u_node_min[a] = bmin[a];
    } else {
      const uint32_t p = u_parent[i];
      const unsigned dgt = level <= PCV_MAX_KEY_LEVELS
                               ? (unsigned)(h_prefix[i] >> (3 * (PCV_MAX_KEY_LEVELS - level))) & 7u
                               : (unsigned)(h_prefix_lo[i] >> (3 * (2 * PCV_MAX_KEY_LEVELS - level))) & 7u;
      const double e = lv.edge[level];
      u_node_min[3 * (size_t)i + 0] = u_node_min[3 * (size_t)p + 0] + (double)((dgt >> 2) & 1) * e;
      u_node_min[3 * (size_t)i + 1] = u_node_min[3 * (size_t)p + 1] + (double)((dgt >> 1) & 1) * e;
      u_node_min[3 * (size_t)i + 2] = u_node_min[3 * (size_t)p + 2] + (double)(dgt & 1) * e;
    }
    const uint64_t np = i == 0 ? pre[0] : pre[i] - ceil8(pre[i]);
    pcv_node_info& ni = t->nodes[i];
    // u128 NodeId = level << 120 | index (node.rs:108-111); the index is the octal path, 3 bits per level
    unsigned __int128 index = 0;
    if (level > PCV_MAX_KEY_LEVELS)
      index = ((unsigned __int128)h_prefix[i] << (3 * (level - PCV_MAX_KEY_LEVELS))) |
              (h_prefix_lo[i] >> (3 * (2 * PCV_MAX_KEY_LEVELS - level)));
    else if (level)
      index = h_prefix[i] >> (3 * (PCV_MAX_KEY_LEVELS - level));
    ni.id_high = ((uint64_t)level << 56) | (uint64_t)(index >> 64);
    ni.id_low = (uint64_t)index;
    ni.num_points = (int64_t)np;
    ni.level = (uint32_t)level;
    ni.encoding = lv.enc[level];
    for (int a = 0; a < 3; ++a) ni.cube_min[a] = u_node_min[3 * (size_t)i + a];
    ni.cube_edge = lv.edge[level];
    ni.xyz_offset = xyz_off;
    ni.point_offset = point_off;
    u_xyz_off[i] = xyz_off;
    u_point_off[i] = point_off;
    xyz_off += (np * 3 * (uint64_t)pcv_bytes_per_coordinate(ni.encoding) + 15) & ~15ull;
    point_off += np;
  }
  t->num_points = point_off;
  t->xyz_bytes = xyz_off;
  t->rgb_bytes = point_off * 3;
  t->int_bytes = t->has_intensity ? point_off * 4 : 0;

  // device tables: walk records for K5, 64-byte node / leaf records for K6 (one contiguous upload)
  uint8_t* rec_base = up + ((up_bytes + 255) & ~(size_t)255);
  PcvNodeRec* u_node_rec = (PcvNodeRec*)rec_base;
  PcvNodeRec* u_leaf_rec = u_node_rec + M;
  for (uint32_t i = 0; i < M; ++i) {
    PcvNodeRec& nr = u_node_rec[i];
    nr.lo = h_lo[i];
    nr.parent = u_parent[i];
    nr.child_off = u_child_off[i];
    nr.enc = lv.enc[u_level[i]];
    nr.edge = lv.edge[u_level[i]];
    // 0 = "no unchecked exact division here": also when the root cube's min is not tame (PcvLevels::fast_ok)
    nr.inv_edge = lv.fast_ok ? lv.inv_edge[u_level[i]] : 0.0;
    nr.inv_edge_lo = lv.fast_ok ? lv.inv_edge_lo[u_level[i]] : 0.0;
    nr.xyz_off = u_xyz_off[i];
    nr.point_off = u_point_off[i];
    for (int a = 0; a < 3; ++a) nr.mn[a] = u_node_min[3 * (size_t)i + a];
  }
  for (uint32_t r = 0; r < num_leaves; ++r) u_leaf_rec[r] = u_node_rec[leaves[r]];
  // climbers of K6 (every 8th point of every leaf; the root is never a leaf): dense index = climb_base[leaf] + j / 8,
  // and the work lists of the leaf-wise settle / climb kernels (pcv_spec.h; PCV_SETTLE_BY_LEAF=0: the slot-wise settle
  // kernel and the flat climb launch, experiments)
  static const bool by_leaf = [] {
    const char* e = pcv_experiment("PCV_SETTLE_BY_LEAF");
    return !e || atoi(e) != 0;
  }();
  const bool fuse_sort = bs->spec && bs->sort_second.pending && (bs->sort_second.nbits <= 7 || !t->has_intensity) && by_leaf && !wide && bs->spec_wide;
  std::vector<uint8_t> fused_leaf;
  uint64_t settled_points = 0;
  uint32_t* u_climb_base = (uint32_t*)(u_leaf_rec + num_leaves);
  const size_t items_off = (((size_t)(M + num_leaves) * sizeof(PcvNodeRec) + (size_t)num_leaves * 4) + 15) & ~(size_t)15;
  PcvSettleItem* u_items = (PcvSettleItem*)((uint8_t*)u_node_rec + items_off);
  uint32_t num_items = 0, num_citems = 0;
  uint64_t num_climbers = 0;
  {
    std::vector<uint32_t> cnt(num_leaves);
    std::vector<uint8_t> climbs(num_leaves);
    for (uint32_t r = 0; r < num_leaves; ++r) {
      cnt[r] = h_hi[leaves[r]] - h_lo[leaves[r]];
      climbs[r] = u_leaf_rec[r].parent != 0xffffffffu;
    }
    // the record sort's held-back second pass settles these leaves itself (PcvSortFuse): integer codes, not the root, no chain
    // to continue; `settle` gets items for the others only (replayed leaves: the pass was queued unfused in front of the replay)
    if (fuse_sort) {
      fused_leaf.assign(num_leaves, 0);
      std::vector<uint8_t> cont_leaf(num_leaves, 0);
      for (uint32_t leaf : bs->cont_nodes) cont_leaf[rank_of[leaf]] = 1;
      for (const auto& fr : bs->fix_ranges) {  // replayed leaves: their records get their codes after the sort (queue_replay)
        uint32_t r = (uint32_t)(std::lower_bound(u_leaf_lo, u_leaf_lo + num_leaves, fr.lo) - u_leaf_lo);
        for (; r < num_leaves && u_leaf_lo[r] == fr.lo; ++r) cont_leaf[r] = 1;  // (empty leaves share their neighbour's first slot)
      }
      std::vector<uint32_t> cnt_left(cnt);
      for (uint32_t r = 0; r < num_leaves; ++r)
        if (climbs[r] && !cont_leaf[r] && u_leaf_rec[r].enc <= PCV_ENC_UINT16) {
          fused_leaf[r] = 1;
          cnt_left[r] = 0;
          settled_points += cnt[r];
        }
      num_items = pcv_settle_items(u_leaf_lo, cnt_left.data(), num_leaves, u_items);
    } else if (by_leaf) {
      num_items = pcv_settle_items(u_leaf_lo, cnt.data(), num_leaves, u_items);
    }
    num_climbers = pcv_climb_layout(cnt.data(), climbs.data(), num_leaves, u_climb_base, u_items + num_items, &num_citems);
    if (!by_leaf) num_citems = 0;
  }
  // single-chain build: leaves below a split first candidate continue their chain from the candidate's codes
  // (spec_continue_kernel): one range per leaf (levels + the candidate's cube min) and one item per <= 512 of its slots
  const size_t cont_off = items_off + (size_t)(num_items + num_citems) * sizeof(PcvSettleItem);
  const uint32_t num_cont = bs->spec ? (uint32_t)bs->cont_nodes.size() : 0u;
  uint8_t* u_cont_ranges = (uint8_t*)u_node_rec + cont_off;
  const size_t cont_items_off = cont_off + (((size_t)num_cont * pcv_cont_range_bytes() + 15) & ~(size_t)15);
  PcvSettleItem* u_cont_items = (PcvSettleItem*)((uint8_t*)u_node_rec + cont_items_off);
  uint32_t num_cont_items = 0;
  {
    std::vector<uint32_t> cont_of_rank;
    if (num_cont && by_leaf) cont_of_rank.assign(num_leaves, 0u);
    for (uint32_t k = 0; k < num_cont; ++k) {
      const uint32_t leaf = bs->cont_nodes[k], from = bs->cont_from[k];
      pcv_fill_cont_range(u_cont_ranges + (size_t)k * pcv_cont_range_bytes(), h_level[from], h_level[leaf], u_node_min + 3 * (size_t)from);
      if (by_leaf) cont_of_rank[rank_of[leaf]] = k + 1;
      for (uint64_t b = h_lo[leaf]; b < h_hi[leaf]; b += kPcvSettleTile)
        u_cont_items[num_cont_items++] = PcvSettleItem{k, (uint32_t)b, (uint32_t)std::min<uint64_t>(b + kPcvSettleTile, h_hi[leaf]), 0u};
    }
    if (num_cont && by_leaf)  // the leaf-wise settle kernel continues these leaves' chains itself (it ignores the mark when it
                              // is launched without the ranges)
      for (uint32_t j = 0; j < num_items; ++j) u_items[j].pad = cont_of_rank[u_items[j].rank];
  }
  const size_t walk_bytes = ((size_t)M * 8 + 255) & ~(size_t)255;
  const size_t fused_off = cont_items_off + (size_t)num_cont_items * sizeof(PcvSettleItem);
  if (fuse_sort) std::memcpy((uint8_t*)u_node_rec + fused_off, fused_leaf.data(), num_leaves);
  const size_t rec_bytes = fused_off + (fuse_sort ? ((size_t)num_leaves + 15) & ~(size_t)15 : 0);
  // the tables live in a context-owned block; with the record sort already running they go up on the side stream (the
  // copy would otherwise queue behind the sort and sit, with its hand-over, between the sort and K6)
  if ((rc = ctx->table_dev_reserve(walk_bytes + rec_bytes + 256))) return rc;
  uint8_t* d_up = (uint8_t*)ctx->table_dev;
  const bool up_side = bs->spec && bs->sort_queued;
  hipStream_t up_st = up_side ? ctx->side : st;
  if (!bs->spec) PCV_HIP_CHECK(ctx, hipMemcpyAsync(d_up, u_walk, (size_t)M * 8, hipMemcpyHostToDevice, up_st));  // K5 only
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(d_up + walk_bytes, u_node_rec, rec_bytes, hipMemcpyHostToDevice, up_st));
  if (up_side && (rc = ctx->side_end())) return ctx->fail(rc, "side stream");
  // (the side stream is in order: this join also covers the layout of the record sort's held-back pass, queued there before the
  // tables — one cross-stream wait in front of that pass instead of two, each ~10 us of idle stream)
  if (up_side) bs->sort_second.join_side = false;
  const uint32_t* d_climb_base = (const uint32_t*)(d_up + walk_bytes + (size_t)(M + num_leaves) * sizeof(PcvNodeRec));
  wt.walk = (const uint64_t*)d_up;
  wt.num_nodes = M;
  PcvPromoteTables pt;
  pt.node_rec = (const PcvNodeRec*)(d_up + walk_bytes);
  pt.leaf_rec = pt.node_rec + M;
  ctx->stage_end(PCV_STAGE_TABLE);  // single-chain build: what is left of the table work once the sort has drained

  if (!bs->spec && (rc = queue_record_sort(ctx, bs, t, &wt, num_leaves, wide))) return rc;
  const bool rec_in_a = bs->rec_in_a;
  uint4 *pay_a = (uint4*)bs->pay_a, *pay_b = (uint4*)bs->pay_b;
  PcvSortPayload& pl = bs->pl;
  uint32_t* s_rank = rec_in_a ? (uint32_t*)keys_a : (uint32_t*)keys_b;
  const void* s_pay = rec_in_a ? (const void*)pay_a : (const void*)pay_b;
  uint32_t** s_plane = rec_in_a ? pl.in : pl.out;
  ctx->stage_begin(PCV_STAGE_PROMOTE_ENCODE);

  // ---- K6 promotion + final encode into node-contiguous blobs ----
  void *bx, *br, *bi = nullptr;
  if ((rc = ctx->dev_alloc(&bx, t->xyz_bytes))) return rc;
  t->d_xyz = (uint8_t*)bx;
  if ((rc = ctx->dev_alloc(&br, t->rgb_bytes))) return rc;
  t->d_rgb = (uint8_t*)br;
  if (t->has_intensity) {
    if ((rc = ctx->dev_alloc(&bi, t->int_bytes))) return rc;
    t->d_int = (uint8_t*)bi;
  }
  if (top_nodes) {  // global-size top nodes are only partly filled by this rank: the rest must read as zero
    const uint64_t tx = top_nodes < M ? u_xyz_off[top_nodes] : t->xyz_bytes;
    const uint64_t tp = top_nodes < M ? u_point_off[top_nodes] : t->num_points;
    if (tx) PCV_HIP_CHECK(ctx, hipMemsetAsync(t->d_xyz, 0, tx, st));
    if (tp) PCV_HIP_CHECK(ctx, hipMemsetAsync(t->d_rgb, 0, tp * 3, st));
    if (tp && t->d_int) PCV_HIP_CHECK(ctx, hipMemsetAsync(t->d_int, 0, tp * 4, st));
  }
  // the compact climber records live in the payload buffer the sort left unused (32 B x n / 8 < 16 B x n)
  void* climbers = rec_in_a ? (void*)pay_b : (void*)pay_a;
  if (fuse_sort) {  // that buffer is the held-back pass's SOURCE: the (16-byte) climbers go to the free half of a key buffer — with
                    // an intensity plane that half carries the plane, and the (32-byte) climbers get a block of their own
    const size_t half = ((size_t)n * 4 + 15) & ~(size_t)15;
    climbers = (void*)((uint8_t*)(rec_in_a ? keys_b : keys_a) + half);
    const size_t need = t->has_intensity ? pcv_climber_bytes(num_climbers) : (size_t)(num_climbers + 1) * 16;
    if (t->has_intensity || half + need > (uint64_t)n * 8) {
      if ((rc = ctx->dev_alloc(&climbers, need))) return rc;
      sc.ptrs.push_back(climbers);
    }
  } else if (pcv_climber_bytes(num_climbers) > (size_t)n * (bs->spec_wide ? 8 : 16)) {
    if ((rc = ctx->dev_alloc(&climbers, pcv_climber_bytes(num_climbers)))) return rc;
    sc.ptrs.push_back(climbers);
  }
  t->settled_in_sort = fuse_sort ? settled_points : 0;
  if (bs->sort_second.pending) {  // the second pass of the record sort, now that tables and blobs exist
    PcvSortFuse fz;
    fz.leaf_rec = pt.leaf_rec;
    fz.leaf_fused = d_up + walk_bytes + fused_off;
    fz.climb_base = d_climb_base;
    fz.climbers = climbers;
    fz.xyz_blob = t->d_xyz;
    fz.rgb_blob = t->d_rgb;
    fz.inten_blob = t->d_int;
    fz.num_leaves = num_leaves;
    ctx->stage_begin(PCV_STAGE_SORT_SECOND);  // (nested in PROMOTE_ENCODE: the sort's pass is accounted on its own)
    rc = pcv_radix_sort_records_second(ctx, &bs->sort_second, fuse_sort ? &fz : nullptr);
    ctx->stage_end(PCV_STAGE_SORT_SECOND);
    if (rc) return rc;
    if ((rc = queue_replay(ctx, bs))) return rc;  // (held back with the pass: replayed leaves rewrite their sorted records)
  }
  // leaves below a split first candidate: the leaf-wise settle kernel continues their chain itself (its items name the
  // range); the slot-wise kernel (experiments) gets the codes rewritten by a pass of its own first
  static const bool fuse_cont = [] {  // PCV_CONT_IN_SETTLE=0 (libpcv_hip_exp.so): the stand-alone continuation kernel
    const char* e = pcv_experiment("PCV_CONT_IN_SETTLE");
    return !e || atoi(e) != 0;
  }();
  const bool cont_in_settle = by_leaf && fuse_cont && num_cont_items > 0;
  if (num_cont_items && !cont_in_settle)
    pcv_launch_spec_continue(ctx, lv, d_up + walk_bytes + cont_off, (const PcvSettleItem*)(d_up + walk_bytes + cont_items_off), num_cont_items,
                             (void*)s_pay, bs->spec_wide);
  pcv_launch_promote_encode(ctx, lv, pt, n, s_rank, s_pay, wide ? s_plane[w_hi] : nullptr,
                            wide ? s_plane[w_hi + 1] : nullptr, wide ? s_plane[w_hi + 2] : nullptr,
                            w_int >= 0 ? s_plane[w_int] : nullptr, d_climb_base, (uint32_t)num_climbers, climbers, t->d_xyz,
                            t->d_rgb, t->d_int, bs->spec_wide,
                            by_leaf ? (const PcvSettleItem*)(d_up + walk_bytes + items_off) : nullptr, num_items,
                            by_leaf ? (const PcvSettleItem*)(d_up + walk_bytes + items_off) + num_items : nullptr, num_citems,
                            cont_in_settle ? (const void*)(d_up + walk_bytes + cont_off) : nullptr);
  ctx->stage_end(PCV_STAGE_PROMOTE_ENCODE);
  PCV_HIP_CHECK(ctx, hipEventRecord(ctx->ev[8], st));
  PCV_HIP_CHECK(ctx, hipGetLastError());
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(st));
  if (bs->spec && bs->resolve_on_device) {
    // the record sort ran on the device's rank map, the tables above come from the host's resolve: both apply the same
    // integer rules to the same counts, so a difference is a bug — reported, never handed out as an octree
    uint32_t dev[2];
    std::memcpy(dev, ctx->mailbox + kMailboxResolve, sizeof(dev));
    if (dev[0] == 0xffffffffu && dev[1] == 0xffffffffu) {  // the sentinel: the resolve kernel never delivered a verdict
      t->nodes.clear();
      t->num_points = 0;
      return ctx->fail(PCV_E_HIP, "single-chain build: the device's resolve kernel wrote no verdict");
    }
    bool same = dev[0] == bs->resolve_host_leaves && dev[1] == 0u;
    if (same && bs->resolve_check_map && !bs->resolve_host_map.empty()) {
      std::vector<uint32_t> dm(bs->resolve_host_map.size());
      PCV_HIP_CHECK(ctx, hipMemcpy(dm.data(), bs->spec_map_dev, dm.size() * 4, hipMemcpyDeviceToHost));
      for (size_t k = 0; same && k < dm.size(); ++k) same = bs->resolve_host_inner[k] || dm[k] == bs->resolve_host_map[k];
    }
    if (!same) {
      t->nodes.clear();
      t->num_points = 0;
      return ctx->fail(PCV_E_HIP, "single-chain build: the device's resolve of the predicted tree disagrees with the host's");
    }
  }
  ctx->prof_resolve();
  for (int sidx = 0; sidx < PCV_STAGE_TOTAL; ++sidx) {
    t->stage_ms[sidx] = 0.f;
    if (ctx->stage_on[sidx]) (void)hipEventElapsedTime(&t->stage_ms[sidx], ctx->stage_b[sidx], ctx->stage_e[sidx]);
  }
  (void)hipEventElapsedTime(&t->stage_ms[PCV_STAGE_TOTAL], ctx->ev[0], ctx->ev[8]);
  return PCV_OK;
}

extern "C" int pcv_build_octree(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points,
                                pcv_octree** out) {
  if (!ctx) return PCV_E_INVALID;
  if (!out) return ctx->fail(PCV_E_INVALID, "out is null");
  pcv_octree* t = nullptr;
  int rc = pcv_build_begin(ctx, params, points, &t);
  if (rc == PCV_OK && (rc = pcv_build_finish(t, nullptr)) != PCV_OK) {
    pcv_octree_free(t);
    t = nullptr;
  }
  *out = t;
  return rc;
}

// ------------------------------------------------------------------------------------------------
// stage-level entry points
// ------------------------------------------------------------------------------------------------
extern "C" int pcv_aabb_reduce(pcv_ctx* ctx, const pcv_points* points, double bbox_min[3], double bbox_max[3]) {
  if (!ctx) return PCV_E_INVALID;
  int rc = validate_points(ctx, points, false);
  if (rc) return rc;
  if (!bbox_min || !bbox_max) return ctx->fail(PCV_E_INVALID, "null output");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  DevPoints d;
  if ((rc = stage_points(ctx, sc, points, false, &d))) return rc;
  return device_aabb(ctx, sc, d, bbox_min, bbox_max);
}

extern "C" int pcv_chain_keys(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points, int nlevels,
                              uint64_t* keys) {
  if (!ctx) return PCV_E_INVALID;
  int rc = validate_points(ctx, points, false);
  if (rc) return rc;
  if (!params || !keys) return ctx->fail(PCV_E_INVALID, "null argument");
  if (!(params->resolution > 0.0)) return ctx->fail(PCV_E_INVALID, "resolution must be positive");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  DevPoints d;
  if ((rc = stage_points(ctx, sc, points, false, &d))) return rc;
  PcvLevels lv;
  int max_level;
  pcv_make_levels(params->bbox_min, params->bbox_max, params->resolution, 64, &lv, &max_level, nullptr, nullptr);
  if (nlevels > 0 && nlevels < lv.nlevels) lv.nlevels = nlevels;
  if (points->n == 0) return PCV_OK;
  uint64_t* dk = keys;
  if (points->mem == PCV_MEM_HOST && (rc = sc.get(&dk, points->n))) return rc;
  pcv_launch_chain_keys(ctx, lv, points->n, 1, d.x, d.y, d.z, dk, false);
  PCV_HIP_CHECK(ctx, hipGetLastError());
  if (points->mem == PCV_MEM_HOST)
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(keys, dk, points->n * 8, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PCV_OK;
}

template <typename KeyT>
static int sort_api(pcv_ctx* ctx, KeyT* keys, uint32_t* values, uint64_t n, int begin_bit, int end_bit, int mem) {
  if (!ctx) return PCV_E_INVALID;
  if (n == 0) return PCV_OK;
  if (!keys) return ctx->fail(PCV_E_INVALID, "keys is null");
  if (begin_bit < 0 || end_bit > (int)sizeof(KeyT) * 8 || begin_bit > end_bit) return ctx->fail(PCV_E_INVALID, "bad bit range");
  if (n >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "n must be < 2^32 - 1");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  KeyT *a, *b;
  uint32_t *va = nullptr, *vb = nullptr;
  void* scratch;
  int rc;
  if ((rc = sc.get(&a, n)) || (rc = sc.get(&b, n))) return rc;
  if (values && ((rc = sc.get(&va, n)) || (rc = sc.get(&vb, n)))) return rc;
  if ((rc = ctx->dev_alloc(&scratch, pcv_sort_scratch_bytes(n)))) return rc;
  sc.ptrs.push_back(scratch);
  hipMemcpyKind in = mem == PCV_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  hipMemcpyKind outk = mem == PCV_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(a, keys, n * sizeof(KeyT), in, ctx->stream));
  if (values) PCV_HIP_CHECK(ctx, hipMemcpyAsync(va, values, n * 4, in, ctx->stream));
  PcvSortPayload pl;
  pl.nwords = values ? 1 : 0;
  pl.in[0] = va;
  pl.out[0] = vb;
  bool in_a = true;
  if constexpr (sizeof(KeyT) == 8) rc = pcv_radix_sort_u64(ctx, (uint64_t*)a, (uint64_t*)b, n, begin_bit, end_bit, &pl, scratch, &in_a);
  else rc = pcv_radix_sort_u32(ctx, (uint32_t*)a, (uint32_t*)b, n, begin_bit, end_bit, &pl, scratch, &in_a);
  if (rc) return rc;
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(keys, in_a ? a : b, n * sizeof(KeyT), outk, ctx->stream));
  if (values) PCV_HIP_CHECK(ctx, hipMemcpyAsync(values, in_a ? va : vb, n * 4, outk, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return PCV_OK;
}

extern "C" int pcv_sort_keys64(pcv_ctx* ctx, uint64_t* keys, uint64_t n, int begin_bit, int end_bit, int mem) {
  return sort_api<uint64_t>(ctx, keys, nullptr, n, begin_bit, end_bit, mem);
}
extern "C" int pcv_sort_keys32(pcv_ctx* ctx, uint32_t* keys, uint64_t n, int begin_bit, int end_bit, int mem) {
  return sort_api<uint32_t>(ctx, keys, nullptr, n, begin_bit, end_bit, mem);
}
extern "C" int pcv_sort_pairs32(pcv_ctx* ctx, uint32_t* keys, uint32_t* values, uint64_t n, int begin_bit, int end_bit,
                                int mem) {
  if (ctx && !values) return ctx->fail(PCV_E_INVALID, "values is null");
  return sort_api<uint32_t>(ctx, keys, values, n, begin_bit, end_bit, mem);
}

// ------------------------------------------------------------------------------------------------
// stage-level entry points of the topology / promotion / encode stages
// ------------------------------------------------------------------------------------------------
extern "C" int pcv_node_split(pcv_ctx* ctx, const pcv_build_params* params, const uint64_t* sorted_keys, uint64_t n, int mem,
                              pcv_split_node* nodes, uint64_t capacity, uint64_t* num_nodes) {
  if (!ctx) return PCV_E_INVALID;
  if (!params || !num_nodes || (capacity && !nodes)) return ctx->fail(PCV_E_INVALID, "null argument");
  if (mem != PCV_MEM_HOST && mem != PCV_MEM_DEVICE) return ctx->fail(PCV_E_INVALID, "bad mem");
  if (!(params->resolution > 0.0)) return ctx->fail(PCV_E_INVALID, "resolution must be positive");
  if (n >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "at most 2^32 - 2 keys per call");
  *num_nodes = 0;
  if (n == 0) return PCV_OK;
  if (!sorted_keys) return ctx->fail(PCV_E_INVALID, "sorted_keys is null");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint32_t max_points = params->max_points_per_node ? params->max_points_per_node : PCV_DEFAULT_MAX_POINTS_PER_NODE;
  PcvLevels lv;
  int max_level = 0;
  pcv_make_levels(params->bbox_min, params->bbox_max, params->resolution, 64, &lv, &max_level, nullptr, nullptr);
  PcvScratch sc(ctx);
  int rc;
  const uint64_t* dk = sorted_keys;
  if (mem == PCV_MEM_HOST) {
    uint64_t* tmp;
    if ((rc = sc.get(&tmp, n))) return rc;
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(tmp, sorted_keys, n * 8, hipMemcpyHostToDevice, st));
    dk = tmp;
  }
  PcvNodeTableDev nt;
  uint64_t cap64 = 8ull * (uint64_t)(lv.nlevels + 1) * (n / max_points + 1) + 64;
  if (cap64 > (1ull << 26)) cap64 = 1ull << 26;
  nt.capacity = (uint32_t)cap64;
  nt.prefix_lo = nullptr;
  if ((rc = sc.get(&nt.prefix, cap64)) || (rc = sc.get(&nt.lo, cap64)) || (rc = sc.get(&nt.hi, cap64)) ||
      (rc = sc.get(&nt.parent, cap64)) || (rc = sc.get(&nt.first_child, cap64)) || (rc = sc.get(&nt.level, cap64)) ||
      (rc = sc.get(&nt.child_mask, cap64)) || (rc = sc.get(&nt.open, cap64)) || (rc = sc.get(&nt.counters, 64)))
    return rc;
  nt.max_open = (uint32_t)std::min<uint64_t>(n / max_points + 64, 1u << 22);
  if ((rc = sc.get(&nt.bounds, std::max((size_t)cap64 * 9, (size_t)nt.max_open * 85)))) return rc;
  uint8_t* d_pack;
  if ((rc = sc.get(&d_pack, kPcvPackHeader + ((size_t)cap64 + 8) * sizeof(PcvPackedNode)))) return rc;
  pcv_launch_node_split(ctx, nt, dk, false, (uint32_t)n, lv, params->resolution, max_points, (params->flags >> 8) & 0xffu);
  pcv_launch_pack_node_table(ctx, nt, d_pack);
  PCV_HIP_CHECK(ctx, hipGetLastError());
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->mailbox, d_pack, 256, hipMemcpyDeviceToHost, st));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(st));
  uint32_t counters[64];
  std::memcpy(counters, ctx->mailbox, sizeof(counters));
  if (counters[1] & 2u) return ctx->fail(PCV_E_OOM, "node table capacity exceeded");
  if (counters[1] & 1u) return ctx->fail(PCV_E_DEPTH, "a node at the last key level would still have to be split");
  const uint32_t m = counters[0];
  *num_nodes = m;
  std::vector<PcvPackedNode> pk(m);
  if (m) PCV_HIP_CHECK(ctx, hipMemcpy(pk.data(), d_pack + kPcvPackHeader, (size_t)m * sizeof(PcvPackedNode), hipMemcpyDeviceToHost));
  std::vector<uint32_t> parent(m, 0xffffffffu);
  for (uint32_t i = 0; i < m; ++i)
    if (pk[i].open) {
      const uint32_t nchild = (uint32_t)__builtin_popcount(pk[i].child_mask);
      for (uint32_t c = 0; c < nchild; ++c) parent[pk[i].first_child + c] = i;
    }
  for (uint32_t i = 0; i < m && i < capacity; ++i) {
    pcv_split_node& o = nodes[i];
    const int level = pk[i].level;
    const unsigned __int128 index = level ? (unsigned __int128)(pk[i].prefix >> (3 * (PCV_MAX_KEY_LEVELS - level))) : 0;
    o.id_high = ((uint64_t)level << 56) | (uint64_t)(index >> 64);
    o.id_low = (uint64_t)index;
    o.first = pk[i].lo;
    o.count = (uint64_t)pk[i].hi - pk[i].lo;
    o.level = (uint32_t)level;
    o.parent = parent[i];
    o.first_child = pk[i].open ? pk[i].first_child : 0u;
    o.child_mask = pk[i].child_mask;
    o.is_leaf = pk[i].open ? 0u : 1u;
    o.reserved = 0;
  }
  return PCV_OK;
}

extern "C" int pcv_promote_assign(const pcv_split_node* nodes, uint64_t num_nodes, pcv_promote_node* per_node, uint64_t n,
                                  uint32_t* node_of_slot, uint32_t* slot_in_node) {
  if ((num_nodes && (!nodes || !per_node)) || ((node_of_slot == nullptr) != (slot_in_node == nullptr))) return PCV_E_INVALID;
  if (num_nodes == 0) return PCV_OK;
  if (num_nodes > 0xfffffffeull) return PCV_E_INVALID;
  const uint32_t m = (uint32_t)num_nodes;
  for (uint32_t i = 0; i < m; ++i) {  // children must follow their parent (breadth-first table) and exist
    if (nodes[i].is_leaf) continue;
    const uint32_t nchild = (uint32_t)__builtin_popcount(nodes[i].child_mask & 0xffu);
    if (nchild == 0 || nodes[i].first_child <= i || (uint64_t)nodes[i].first_child + nchild > m) return PCV_E_INVALID;
  }
  // bottom-up stream lengths: |pre(inner)| = sum ceil(|pre(child)| / 8)
  for (uint32_t i = m; i-- > 0;) {
    if (nodes[i].is_leaf) {
      per_node[i].stream_len = nodes[i].count;
    } else {
      uint64_t acc = 0;
      const uint32_t nchild = (uint32_t)__builtin_popcount(nodes[i].child_mask & 0xffu);
      for (uint32_t c = 0; c < nchild; ++c) {
        per_node[nodes[i].first_child + c].child_offset = acc;
        acc += ceil8(per_node[nodes[i].first_child + c].stream_len);
      }
      per_node[i].stream_len = acc;
    }
  }
  per_node[0].child_offset = 0;
  for (uint32_t i = 0; i < m; ++i)
    per_node[i].num_points = i == 0 ? per_node[0].stream_len : per_node[i].stream_len - ceil8(per_node[i].stream_len);
  if (!node_of_slot) return PCV_OK;
  for (uint32_t i = 0; i < m; ++i) {
    if (!nodes[i].is_leaf) continue;
    if (nodes[i].first + nodes[i].count > n) return PCV_E_INVALID;
    for (uint64_t j0 = 0; j0 < nodes[i].count; ++j0) {
      uint32_t node = i;
      uint64_t j = j0;
      while (node != 0 && (j & 7u) == 0) {  // an every-8th element of its stream climbs (generation.rs:222-238)
        j = per_node[node].child_offset + (j >> 3);
        node = nodes[node].parent;
        if (node >= m) return PCV_E_INVALID;
      }
      node_of_slot[nodes[i].first + j0] = node;
      slot_in_node[nodes[i].first + j0] = (uint32_t)(node == 0 ? j : j - (j >> 3) - 1);
    }
  }
  return PCV_OK;
}

extern "C" int pcv_gather_encode(pcv_ctx* ctx, const pcv_build_params* params, const pcv_points* points,
                                 const pcv_split_node* nodes, uint64_t num_nodes, pcv_octree** out) {
  if (!ctx) return PCV_E_INVALID;
  if (!out) return ctx->fail(PCV_E_INVALID, "out is null");
  *out = nullptr;
  if (!params) return ctx->fail(PCV_E_INVALID, "params is null");
  int rc = validate_points(ctx, points, true);
  if (rc) return rc;
  if (params->flags & PCV_BUILD_COMPUTE_BBOX) return ctx->fail(PCV_E_INVALID, "the topology was built for a given bounding box: pass it");
  if (points->n && (!nodes || num_nodes == 0)) return ctx->fail(PCV_E_INVALID, "no topology");
  if (num_nodes > (1ull << 26)) return ctx->fail(PCV_E_INVALID, "too many nodes");
  PcvTrueTree tt;
  const uint32_t m = (uint32_t)num_nodes;
  for (uint32_t i = 0; i < m && points->n; ++i) {
    const pcv_split_node& nd = nodes[i];
    if (nd.level > PCV_MAX_KEY_LEVELS) return ctx->fail(PCV_E_INVALID, "pcv_gather_encode takes trees of up to 21 levels");
    if (nd.first + nd.count > points->n) return ctx->fail(PCV_E_INVALID, "node range outside the points");
    const uint32_t nchild = (uint32_t)__builtin_popcount(nd.child_mask & 0xffu);
    if (!nd.is_leaf && (nchild == 0 || nd.first_child <= i || (uint64_t)nd.first_child + nchild > m))
      return ctx->fail(PCV_E_INVALID, "node table is not breadth first with consecutive children");
    const unsigned __int128 index = ((unsigned __int128)(nd.id_high & 0x00ffffffffffffffull) << 64) | nd.id_low;
    tt.prefix.push_back(nd.level ? (uint64_t)(index << (3 * (PCV_MAX_KEY_LEVELS - nd.level))) : 0ull);
    tt.lo.push_back((uint32_t)nd.first);
    tt.hi.push_back((uint32_t)(nd.first + nd.count));
    tt.first_child.push_back(nd.is_leaf ? 0u : nd.first_child);
    tt.level.push_back((uint8_t)nd.level);
    tt.child_mask.push_back((uint8_t)nd.child_mask);
    tt.open.push_back(nd.is_leaf ? 0 : 1);
  }
  if (points->n && (tt.lo[0] != 0 || tt.hi[0] != points->n || tt.level[0] != 0))
    return ctx->fail(PCV_E_INVALID, "the first node must be the root and span all points");
  pcv_octree* t = nullptr;
  rc = build_begin_impl(ctx, params, points, nullptr, &t, &tt);
  if (rc == PCV_OK && (rc = pcv_build_finish(t, nullptr)) != PCV_OK) {
    pcv_octree_free(t);
    t = nullptr;
  }
  *out = t;
  return rc;
}
