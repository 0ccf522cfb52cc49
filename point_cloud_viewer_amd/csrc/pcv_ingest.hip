// pcv_ingest.hip — streaming batch ingest in the reference's own layout.
//
// The reference's build entry takes `impl Iterator<Item = PointsBatch>` (src/octree/generation.rs:289-295); a PointsBatch
// holds `position: Vec<Point3<f64>>` — AoS, 24 bytes per point — plus "color" `Vec<Vector3<u8>>` and optionally "intensity"
// `Vec<f32>` (src/lib.rs:102-107, src/octree/mod.rs:62-74), 500 000 points at a time (src/lib.rs:52). This file is the
// boundary for exactly that stream:
//
//   pcv_ingest_begin   device SoA arrays sized from NumberOfPoints::num_points (a hint: they grow if the stream is longer)
//   pcv_ingest_append  one batch AS IT IS: the three arrays are copied side by side into one chunk of the context's ring of
//                      pinned chunks (host threads), ONE DMA carries the chunk to a device staging chunk, and ONE kernel
//                      transposes the positions AoS -> SoA through LDS into their place behind the points already there,
//                      copies colour / intensity behind theirs, and folds the batch into the running bounding box
//                      (find_bounding_box, generation.rs:256-270 — Aabb::grow, aabb.rs:41-44). The call returns as soon as
//                      the DMA and the kernel are queued: the producer decodes its next batch while this one goes up.
//   pcv_ingest_finish  pcv_build_octree on the device-resident cloud (with PCV_BUILD_COMPUTE_BBOX: the box folded during the
//                      ingest — no pass over the cloud), then the arrays go back to the pool.
//
// Host memory is O(batch) (the ring: 3 x 32 MiB), never O(cloud); nothing is transposed on the host.
#include <algorithm>
#include <cstring>
#include <limits>

#include "pcv_internal.h"
#include <chrono>
#include <vector>

namespace {

constexpr int kIngestBlock = 256;
constexpr int kIngestTile = 1024;  // points per tile: 24 KB of LDS
// points of one DMA: 24 + 3 + 4 bytes each plus three 16-byte alignment gaps must fit a ring chunk (32 MiB)
constexpr uint64_t kIngestSub = 1u << 20;
static_assert(kIngestSub * 31 + 64 <= pcv_ctx::kRingChunk, "a sub-batch must fit one ring chunk");

// f64 -> u64 whose unsigned order is the order of the doubles (-0.0 < +0.0, as v_min_f64 / v_max_f64 order them)
__device__ __forceinline__ unsigned long long ordered_key(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
inline double ordered_value(uint64_t k) {
  const uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double v;
  std::memcpy(&v, &u, 8);
  return v;
}
inline uint64_t ordered_key_host(double v) {
  uint64_t u;
  std::memcpy(&u, &v, 8);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ double wave_fmin(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_fmax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// dst and src have the SAME alignment modulo 16 (the host lays the chunk out that way): head bytes, 16-byte body, tail bytes
__device__ __forceinline__ void copy_congruent16(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t bytes, uint32_t tid,
                                                 uint32_t nthreads) {
  uint32_t head = (16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u;
  if (head > bytes) head = bytes;
  const uint32_t body = (bytes - head) >> 4;
  const uint32_t tail = bytes - head - (body << 4);
  if (tid < head) dst[tid] = src[tid];
  const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src + head);
  uint4* __restrict__ d4 = reinterpret_cast<uint4*>(dst + head);
  for (uint32_t i = tid; i < body; i += nthreads) d4[i] = s4[i];
  if (tid < tail) dst[head + (body << 4) + tid] = src[head + (body << 4) + tid];
}

// One batch: positions AoS (n x 3 f64 at `aos`, 16-byte aligned) -> x / y / z (already offset to the batch's first slot);
// colour and intensity bytes behind the ones already there; the batch's min / max folded into acc[6] (ordered keys).
__global__ __launch_bounds__(kIngestBlock) void ingest_batch_kernel(uint32_t n, const double* __restrict__ aos, double* __restrict__ x,
                                                                     double* __restrict__ y, double* __restrict__ z,
                                                                     const uint8_t* __restrict__ src_rgb, uint8_t* __restrict__ dst_rgb,
                                                                     const uint8_t* __restrict__ src_int, uint8_t* __restrict__ dst_int,
                                                                     unsigned long long* __restrict__ acc) {
  __shared__ double tile[kIngestTile * 3];
  __shared__ double red[kIngestBlock / 64][6];
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
  const uint32_t tiles = (n + kIngestTile - 1) / kIngestTile;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t first = t * kIngestTile;
    const uint32_t cnt = n - first < (uint32_t)kIngestTile ? n - first : (uint32_t)kIngestTile;
    const uint32_t words = cnt * 3;  // doubles of this tile; its first double sits at an even index: 16-byte loads are aligned
    const double* __restrict__ in = aos + (uint64_t)first * 3;
#pragma unroll
    for (int j = 0; j < kIngestTile * 3 / (2 * kIngestBlock); ++j) {
      const uint32_t w = 2 * (j * kIngestBlock + threadIdx.x);
      if (w + 1 < words) {
        const double2 v = *reinterpret_cast<const double2*>(in + w);
        tile[w] = v.x;
        tile[w + 1] = v.y;
      } else if (w < words) {
        tile[w] = in[w];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kIngestTile / kIngestBlock; ++q) {
      const uint32_t p = q * kIngestBlock + threadIdx.x;
      if (p < cnt) {
        const double vx = tile[3 * p], vy = tile[3 * p + 1], vz = tile[3 * p + 2];
        x[first + p] = vx;
        y[first + p] = vy;
        z[first + p] = vz;
        lo[0] = fmin(lo[0], vx), hi[0] = fmax(hi[0], vx);
        lo[1] = fmin(lo[1], vy), hi[1] = fmax(hi[1], vy);
        lo[2] = fmin(lo[2], vz), hi[2] = fmax(hi[2], vz);
      }
    }
    __syncthreads();
  }
  const uint32_t gtid = blockIdx.x * kIngestBlock + threadIdx.x, gthreads = gridDim.x * kIngestBlock;
  copy_congruent16(dst_rgb, src_rgb, n * 3u, gtid, gthreads);
  if (dst_int) copy_congruent16(dst_int, src_int, n * 4u, gtid, gthreads);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double l = wave_fmin(lo[a]), h = wave_fmax(hi[a]);
    if (lane == 0) {
      red[wave][a] = l;
      red[wave][3 + a] = h;
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = red[0][threadIdx.x];
    for (int w = 1; w < kIngestBlock / 64; ++w) v = threadIdx.x < 3 ? fmin(v, red[w][threadIdx.x]) : fmax(v, red[w][threadIdx.x]);
    // a block that saw no point (or only NaNs: fmin / fmax skip them like K1 does) leaves the accumulator alone
    if (threadIdx.x < 3) {
      if (v < inf) atomicMin(&acc[threadIdx.x], ordered_key(v));
    } else {
      if (v > -inf) atomicMax(&acc[threadIdx.x], ordered_key(v));
    }
  }
}

}  // namespace

struct pcv_ingest {
  pcv_ctx* ctx = nullptr;
  uint64_t n = 0, cap = 0;
  bool has_intensity = false;
  double *x = nullptr, *y = nullptr, *z = nullptr;
  uint8_t* rgb = nullptr;
  float* inten = nullptr;
  uint8_t* stage[pcv_ctx::kRingSlots] = {};  // device partners of the pinned ring chunks
  hipEvent_t read_ev[pcv_ctx::kRingSlots] = {};  // the transposition kernel has read the staging chunk: the copy stream may refill it
  bool read_busy[pcv_ctx::kRingSlots] = {};
  // the pinned chunk being filled: batches are packed into it one after the other and go up in ONE DMA when the next batch does
  // not fit (or at finish): a DMA of 13.5 MB — one batch of 500 000 points — runs at 40 GB/s, one of 27 MB at 50
  struct Seg {
    size_t off, rgb_off, int_off;  // byte offsets inside the chunk
    uint32_t m;
    uint64_t at;  // index of the segment's first point in the device arrays
  };
  int cur_slot = -1;
  size_t cur_fill = 0;
  std::vector<Seg> segs;
  unsigned long long* acc = nullptr;         // 6 ordered keys: running min xyz, max xyz
  bool failed = false;
};

static void ingest_release(pcv_ingest* g) {
  pcv_ctx* ctx = g->ctx;
  if (g->cur_slot >= 0 && ctx->ring_held == g->cur_slot) ctx->ring_held = -1;
  for (void* p : {(void*)g->x, (void*)g->y, (void*)g->z, (void*)g->rgb, (void*)g->inten, (void*)g->acc})
    if (p) ctx->dev_free(p);
  for (auto& s : g->stage)
    if (s) ctx->dev_free(s);
  for (auto& e : g->read_ev)
    if (e) (void)hipEventDestroy(e);
  delete g;
}

static int ingest_reserve(pcv_ingest* g, uint64_t want) {
  if (want <= g->cap) return PCV_OK;
  pcv_ctx* ctx = g->ctx;
  uint64_t cap = g->cap + g->cap / 2;
  if (cap < want) cap = want;
  if (cap < (1u << 20)) cap = 1u << 20;
  if (cap > 0xfffffffeull) cap = 0xfffffffeull;
  cap = (cap + 15) & ~(uint64_t)15;
  double *x = nullptr, *y = nullptr, *z = nullptr;
  uint8_t* rgb = nullptr;
  float* inten = nullptr;
  int rc;
  if ((rc = ctx->dev_alloc((void**)&x, cap * 8)) || (rc = ctx->dev_alloc((void**)&y, cap * 8)) || (rc = ctx->dev_alloc((void**)&z, cap * 8)) ||
      (rc = ctx->dev_alloc((void**)&rgb, cap * 3 + 16)) || (g->has_intensity && (rc = ctx->dev_alloc((void**)&inten, cap * 4 + 16)))) {
    for (void* p : {(void*)x, (void*)y, (void*)z, (void*)rgb, (void*)inten})
      if (p) ctx->dev_free(p);
    return rc;
  }
  if (g->n) {  // the stream turned out longer than the hint: move what is there (device to device, stream-ordered)
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(x, g->x, g->n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(y, g->y, g->n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(z, g->z, g->n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    PCV_HIP_CHECK(ctx, hipMemcpyAsync(rgb, g->rgb, g->n * 3, hipMemcpyDeviceToDevice, ctx->stream));
    if (inten) PCV_HIP_CHECK(ctx, hipMemcpyAsync(inten, g->inten, g->n * 4, hipMemcpyDeviceToDevice, ctx->stream));
  }
  for (void* p : {(void*)g->x, (void*)g->y, (void*)g->z, (void*)g->rgb, (void*)g->inten})
    if (p) ctx->dev_free(p);  // the pool is stream-ordered: the copies above are queued before any reuse
  g->x = x, g->y = y, g->z = z, g->rgb = rgb, g->inten = inten;
  g->cap = cap;
  return PCV_OK;
}

extern "C" int pcv_ingest_begin(pcv_ctx* ctx, uint64_t num_points_hint, int has_intensity, pcv_ingest** out) {
  if (!ctx) return PCV_E_INVALID;
  if (!out) return ctx->fail(PCV_E_INVALID, "out is null");
  *out = nullptr;
  if (num_points_hint >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "at most 2^32 - 2 points per build");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc = ctx->ring_ensure();
  if (rc) return rc;
  pcv_ingest* g = new pcv_ingest();
  g->ctx = ctx;
  g->has_intensity = has_intensity != 0;
  if ((rc = ctx->dev_alloc((void**)&g->acc, 64))) {
    ingest_release(g);
    return rc;
  }
  for (auto& s : g->stage)
    if ((rc = ctx->dev_alloc((void**)&s, pcv_ctx::kRingChunk))) {
      ingest_release(g);
      return rc;
    }
  for (auto& e : g->read_ev)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      ingest_release(g);
      return ctx->fail(PCV_E_HIP, "pcv_ingest_begin: hipEventCreate");
    }
  const double inf = std::numeric_limits<double>::infinity();
  uint64_t init[6] = {ordered_key_host(inf), ordered_key_host(inf), ordered_key_host(inf),
                      ordered_key_host(-inf), ordered_key_host(-inf), ordered_key_host(-inf)};
  std::memcpy(ctx->mailbox, init, sizeof(init));  // pinned: the copy below reads it in stream order
  if (hipMemcpyAsync(g->acc, ctx->mailbox, sizeof(init), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess || (rc = ingest_reserve(g, num_points_hint ? num_points_hint : 1))) {
    ingest_release(g);
    return rc ? rc : ctx->fail(PCV_E_HIP, "pcv_ingest_begin: could not initialise the bounding-box accumulator");
  }
  *out = g;
  return PCV_OK;
}

extern "C" uint64_t pcv_ingest_num_points(const pcv_ingest* g) { return g ? g->n : 0; }

// PCV_INGEST_TRACE=1 (libpcv_hip_exp.so): where an append's host time goes, summed over the ingest and printed by finish
static double g_trace_us[4];  // wait for the ring slot, copy into the pinned chunk, queue DMA + kernel, whole call
static inline double trace_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The pinned chunk in hand goes up: one DMA, one transposition kernel per batch in it.
// The DMAs travel on the side stream, the kernels on `stream`: on ONE stream every DMA pays two hand-overs between the copy
// engine and the compute queue (DMA k+1 cannot start before kernel k has ended). The kernels wait for their DMA (ring_ev), the DMA
// that refills a staging chunk for the kernels that read it (read_ev); the stream order of `stream` is what keeps the
// destination arrays (and their growth, ingest_reserve) consistent. PCV_INGEST_ONE_STREAM=1 (libpcv_hip_exp.so): everything on `stream`.
static int ingest_flush(pcv_ingest* g) {
  if (g->cur_slot < 0) return PCV_OK;
  pcv_ctx* ctx = g->ctx;
  const int slot = g->cur_slot;
  g->cur_slot = -1;
  if (ctx->ring_held == slot) ctx->ring_held = -1;
  static const bool one_stream = [] {
    const char* e = pcv_experiment("PCV_INGEST_ONE_STREAM");
    return (e && atoi(e) != 0);
  }();
  hipStream_t cs = one_stream || !ctx->side ? ctx->stream : ctx->side;
  uint8_t* st = g->stage[slot];
  bool ok = true;
  if (cs != ctx->stream && g->read_busy[slot]) ok = hipStreamWaitEvent(cs, g->read_ev[slot], 0) == hipSuccess;
  ok = ok && hipMemcpyAsync(st, ctx->ring[slot], g->cur_fill, hipMemcpyHostToDevice, cs) == hipSuccess &&
       hipEventRecord(ctx->ring_ev[slot], cs) == hipSuccess;
  if (ok && cs != ctx->stream) ok = hipStreamWaitEvent(ctx->stream, ctx->ring_ev[slot], 0) == hipSuccess;
  if (!ok) {
    g->failed = true;
    return ctx->fail(PCV_E_HIP, "pcv_ingest_append: queuing the DMA of a chunk of batches failed");
  }
  ctx->ring_busy[slot] = true;
  for (const pcv_ingest::Seg& sg : g->segs) {
    PcvProf prof(ctx, PCV_K_INGEST);
    const uint32_t tiles = (uint32_t)((sg.m + kIngestTile - 1) / kIngestTile);
    hipLaunchKernelGGL(ingest_batch_kernel, dim3(tiles < 1024u ? tiles : 1024u), dim3(kIngestBlock), 0, ctx->stream, sg.m,
                       (const double*)(st + sg.off), g->x + sg.at, g->y + sg.at, g->z + sg.at, (const uint8_t*)(st + sg.rgb_off), g->rgb + sg.at * 3,
                       (const uint8_t*)(st + sg.int_off), g->has_intensity ? (uint8_t*)(g->inten + sg.at) : nullptr, g->acc);
  }
  g->segs.clear();
  if (hipGetLastError() != hipSuccess || (cs != ctx->stream && hipEventRecord(g->read_ev[slot], ctx->stream) != hipSuccess)) {
    g->failed = true;
    return ctx->fail(PCV_E_HIP, "pcv_ingest_append: launching ingest_batch_kernel failed");
  }
  g->read_busy[slot] = cs != ctx->stream;
  return PCV_OK;
}

extern "C" int pcv_ingest_append(pcv_ingest* g, const double* xyz, const uint8_t* rgb, const float* intensity, uint64_t n) {
  if (!g) return PCV_E_INVALID;
  static const bool trace = pcv_experiment("PCV_INGEST_TRACE") != nullptr;
  const double tr0 = trace ? trace_now() : 0.0;
  pcv_ctx* ctx = g->ctx;
  if (g->failed) return ctx->fail(PCV_E_INVALID, "pcv_ingest_append after a failed append: the ingest can only be finished or aborted");
  if (n == 0) return PCV_OK;
  if (!xyz || !rgb) return ctx->fail(PCV_E_INVALID, "positions and colour are required (on_disk.rs:20-22: colour is always present)");
  if (g->has_intensity && !intensity) return ctx->fail(PCV_E_INVALID, "the ingest was begun with intensity: every batch must carry it (generation.rs:167-177 unwraps)");
  if (g->n + n >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "at most 2^32 - 2 points per build");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc = ingest_reserve(g, g->n + n);
  if (rc) return rc;
  const size_t nworkers = ctx->host_pool.threads.size() + 1;
  for (uint64_t done = 0; done < n; done += kIngestSub) {
    const uint64_t m = n - done < kIngestSub ? n - done : kIngestSub;
    // the batch's place in the chunk: positions at a 256-byte boundary; colour and intensity at offsets CONGRUENT modulo 16 to
    // where they go on the device (the arrays' bases are 256-byte aligned), so that the kernel copies whole 16-byte words
    const size_t xyz_bytes = (size_t)m * 24, rgb_bytes = (size_t)m * 3, int_bytes = g->has_intensity ? (size_t)m * 4 : 0;
    size_t off, rgb_off, int_off, total;
    auto place = [&](size_t base) {
      off = (base + 255) & ~(size_t)255;
      rgb_off = ((off + xyz_bytes + 15) & ~(size_t)15) + ((g->n * 3) & 15);
      int_off = ((rgb_off + rgb_bytes + 15) & ~(size_t)15) + (g->has_intensity ? ((g->n * 4) & 15) : 0);
      total = int_off + int_bytes;
    };
    place(g->cur_slot >= 0 ? g->cur_fill : 0);
    double tr1 = trace ? trace_now() : 0.0, tr2 = tr1;
    if (g->cur_slot >= 0 && total > pcv_ctx::kRingChunk) {  // the chunk in hand is full: up it goes
      if ((rc = ingest_flush(g))) return rc;
      place(0);
      if (trace) g_trace_us[2] += trace_now() - tr1, tr1 = tr2 = trace_now();
    }
    if (g->cur_slot < 0) {
      const int slot = ctx->ring_take();
      if (ctx->ring_busy[slot]) PCV_HIP_CHECK(ctx, hipEventSynchronize(ctx->ring_ev[slot]));  // its previous DMA has left the chunk
      g->cur_slot = slot;
      g->cur_fill = 0;
      if (ctx->ring_held < 0) ctx->ring_held = slot;  // (one ingest fills a chunk at a time: a second ingest on the same context flushes as it goes, below)
      tr2 = trace ? trace_now() : 0.0;
    }
    uint8_t* chunk = (uint8_t*)ctx->ring[g->cur_slot];
    const uint8_t* sx = (const uint8_t*)(xyz + done * 3);
    const uint8_t* sc = rgb + done * 3;
    const uint8_t* si = g->has_intensity ? (const uint8_t*)(intensity + done) : nullptr;
    if (xyz_bytes + rgb_bytes + int_bytes < (256u << 10) || nworkers == 1) {
      std::memcpy(chunk + off, sx, xyz_bytes);
      std::memcpy(chunk + rgb_off, sc, rgb_bytes);
      if (int_bytes) std::memcpy(chunk + int_off, si, int_bytes);
    } else {
      // one part per worker over the three regions laid end to end (the caller's thread works too)
      const size_t flat = xyz_bytes + rgb_bytes + int_bytes;
      const size_t part = std::max<size_t>(128u << 10, ((flat + nworkers - 1) / nworkers + 4095) & ~(size_t)4095);
      ctx->host_pool.run((flat + part - 1) / part, [&](size_t p) {
        size_t b = p * part, e = b + part < flat ? b + part : flat;
        while (b < e) {  // [b, e) may straddle regions
          if (b < xyz_bytes) {
            const size_t len = (e < xyz_bytes ? e : xyz_bytes) - b;
            std::memcpy(chunk + off + b, sx + b, len);
            b += len;
          } else if (b < xyz_bytes + rgb_bytes) {
            const size_t o = b - xyz_bytes, len = (e < xyz_bytes + rgb_bytes ? e : xyz_bytes + rgb_bytes) - b;
            std::memcpy(chunk + rgb_off + o, sc + o, len);
            b += len;
          } else {
            const size_t o = b - xyz_bytes - rgb_bytes, len = e - b;
            std::memcpy(chunk + int_off + o, si + o, len);
            b += len;
          }
        }
      });
    }
    g->segs.push_back(pcv_ingest::Seg{off, rgb_off, int_off, (uint32_t)m, g->n});
    g->cur_fill = total;
    g->n += m;
    if (ctx->ring_held != g->cur_slot && (rc = ingest_flush(g))) return rc;  // another ingest of this context holds the marker
    if (trace) g_trace_us[0] += tr2 - tr1, g_trace_us[1] += trace_now() - tr2;
  }
  if (trace) g_trace_us[3] += trace_now() - tr0;
  return PCV_OK;
}

extern "C" void pcv_ingest_abort(pcv_ingest* g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  if (g->ctx->side) (void)hipStreamSynchronize(g->ctx->side);  // a DMA whose kernel was never queued (a failed append)
  (void)hipStreamSynchronize(g->ctx->stream);
  ingest_release(g);
}

extern "C" int pcv_ingest_bbox(pcv_ingest* g, double bbox_min[3], double bbox_max[3]) {
  if (!g) return PCV_E_INVALID;
  pcv_ctx* ctx = g->ctx;
  if (!bbox_min || !bbox_max) return ctx->fail(PCV_E_INVALID, "null output");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!g->failed) {  // the batches still in the pinned chunk in hand
    const int frc = ingest_flush(g);
    if (frc) return frc;
  }
  PCV_HIP_CHECK(ctx, hipMemcpyAsync(ctx->mailbox, g->acc, 48, hipMemcpyDeviceToHost, ctx->stream));
  PCV_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const double inf = std::numeric_limits<double>::infinity();
  for (int a = 0; a < 3; ++a) {
    bbox_min[a] = ordered_value(ctx->mailbox[a]);
    bbox_max[a] = ordered_value(ctx->mailbox[3 + a]);
    // no (non-NaN) point yet: Aabb::zero() (generation.rs:269), what pcv_aabb_reduce reports for n == 0
    if (g->n == 0 || bbox_min[a] == inf) bbox_min[a] = bbox_max[a] = 0.0;
  }
  return PCV_OK;
}

extern "C" int pcv_ingest_finish(pcv_ingest* g, const pcv_build_params* params, pcv_octree** out) {
  if (pcv_experiment("PCV_INGEST_TRACE")) {
    fprintf(stderr, "[ingest] wait slot %.1f ms, copy to pinned %.1f ms, queue %.1f ms, appends in all %.1f ms\n", g_trace_us[0] * 1e-3,
            g_trace_us[1] * 1e-3, g_trace_us[2] * 1e-3, g_trace_us[3] * 1e-3);
    g_trace_us[0] = g_trace_us[1] = g_trace_us[2] = g_trace_us[3] = 0.0;
  }
  if (!g) return PCV_E_INVALID;
  pcv_ctx* ctx = g->ctx;
  int rc = PCV_OK;
  if (!out || !params)
    rc = ctx->fail(PCV_E_INVALID, "null argument");
  else if (g->failed)
    rc = ctx->fail(PCV_E_INVALID, "pcv_ingest_finish after a failed append");
  if (out) *out = nullptr;
  if (rc == PCV_OK) rc = ingest_flush(g);  // the batches still in the pinned chunk in hand
  if (rc == PCV_OK) {
    pcv_build_params p = *params;
    if (p.flags & PCV_BUILD_COMPUTE_BBOX) {  // the box was folded batch by batch: no pass over the cloud
      rc = pcv_ingest_bbox(g, p.bbox_min, p.bbox_max);
      p.flags &= ~PCV_BUILD_COMPUTE_BBOX;
    }
    if (rc == PCV_OK) {
      pcv_points pts{};
      pts.n = g->n;
      pts.x = g->x, pts.y = g->y, pts.z = g->z;
      pts.color = g->rgb;
      pts.color_stride = 3;
      pts.intensity = g->has_intensity ? g->inten : nullptr;
      pts.mem = PCV_MEM_DEVICE;
      rc = pcv_build_octree(ctx, &p, &pts, out);  // synchronous: the arrays outlive every kernel that reads them
    }
  }
  (void)hipSetDevice(ctx->device);
  if (g->failed && ctx->side) (void)hipStreamSynchronize(ctx->side);  // a DMA whose kernel was never queued
  (void)hipStreamSynchronize(ctx->stream);
  ingest_release(g);
  return rc;
}
