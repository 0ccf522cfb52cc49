// pcv_ply_dev.hip — build_octree_from_file (reference src/octree/generation.rs:272-287) with the PLY decode on the device.
//
// The host-side ingest (pcv_ply.cpp) turns every vertex record into f64 SoA arrays before anything reaches the GPU:
// 27 bytes per point over the link for a file that holds 15 (float x y z + uchar r g b). Here the vertex records go
// up exactly as they are in the file — pread by the context's host threads into the pinned staging ring, one DMA per
// chunk — and one kernel casts x / y / z to f64 and adds the header's `comment offset` exactly like
// batch_from_readers (src/read_write/ply.rs:488-493: `Point3::new(x, y, z) + offset`, f64 arithmetic), splitting colour
// and intensity into the arrays the build reads. Nothing else differs from pcv_build_octree on device-resident points.
#include <fcntl.h>
#include <unistd.h>

#include "pcv_internal.h"
#include "pcv_ply_layout.h"

namespace {

// Unaligned little-endian scalar of a vertex record -> f64 (ply.rs:328-455: every scalar type is read with `as f64`).
__device__ __forceinline__ double ply_scalar_f64(int type, const uint8_t* __restrict__ p) {
  switch (type) {  // kernel-uniform
    case PCV_PLY_I8: return (double)(int8_t)p[0];
    case PCV_PLY_U8: return (double)p[0];
    case PCV_PLY_I16: return (double)(int16_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8));
    case PCV_PLY_U16: return (double)(uint16_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8));
    case PCV_PLY_I32: {
      int32_t v;
      __builtin_memcpy(&v, p, 4);
      return (double)v;
    }
    case PCV_PLY_U32: {
      uint32_t v;
      __builtin_memcpy(&v, p, 4);
      return (double)v;
    }
    case PCV_PLY_F32: {
      float v;
      __builtin_memcpy(&v, p, 4);
      return (double)v;
    }
    default: {
      double v;
      __builtin_memcpy(&v, p, 8);
      return v;
    }
  }
}

struct PlyDecodeArgs {
  int stride;
  int x_type, x_off, y_type, y_off, z_type, z_off, r_off, g_off, b_off, i_off;
  double offset[3];
};

// One point per lane: the record bytes of a wave are one contiguous span (64 x stride bytes), so the byte / dword
// loads of its lanes share cache lines; outputs are coalesced SoA stores.
__global__ __launch_bounds__(256) void ply_decode_kernel(PlyDecodeArgs a, uint64_t n, const uint8_t* __restrict__ raw,
                                                          double* __restrict__ x, double* __restrict__ y, double* __restrict__ z,
                                                          uint8_t* __restrict__ rgb, float* __restrict__ intensity) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t* __restrict__ p = raw + i * (uint64_t)a.stride;
  x[i] = ply_scalar_f64(a.x_type, p + a.x_off) + a.offset[0];
  y[i] = ply_scalar_f64(a.y_type, p + a.y_off) + a.offset[1];
  z[i] = ply_scalar_f64(a.z_type, p + a.z_off) + a.offset[2];
  if (rgb) {
    rgb[3 * i] = p[a.r_off];
    rgb[3 * i + 1] = p[a.g_off];
    rgb[3 * i + 2] = p[a.b_off];
  }
  if (intensity) {
    float f;
    __builtin_memcpy(&f, p + a.i_off, 4);
    intensity[i] = f;
  }
}

}  // namespace

extern "C" int pcv_build_octree_from_ply(pcv_ctx* ctx, const pcv_build_params* params, const char* path, int with_intensity,
                                         pcv_octree** out) {
  if (!ctx) return PCV_E_INVALID;
  if (!out) return ctx->fail(PCV_E_INVALID, "out is null");
  *out = nullptr;
  if (!params || !path) return ctx->fail(PCV_E_INVALID, "null argument");
  FILE* f = fopen(path, "rb");
  if (!f) return ctx->fail(PCV_E_IO, "Could not open input file.");
  PcvPlyLayout lay;
  char err[256] = {0};
  int rc = pcv_ply_parse_header(f, &lay, err, sizeof(err));
  fclose(f);
  if (rc != PCV_OK) return ctx->fail(rc, err);
  if (lay.r_off < 0) return ctx->fail(PCV_E_INVALID, "the PLY has no red/green/blue properties; the octree format requires colour");
  if (with_intensity && lay.i_off < 0) return ctx->fail(PCV_E_INVALID, "attribute 'intensity' requested but the PLY has none");
  const uint64_t n = (uint64_t)lay.vertex_count;
  if (n >= 0xffffffffull) return ctx->fail(PCV_E_INVALID, "at most 2^32 - 2 points per call");
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PcvScratch sc(ctx);
  pcv_points pts{};
  pts.n = n;
  pts.mem = PCV_MEM_DEVICE;
  pts.color_stride = 3;
  if (n > 0) {
    const size_t raw_bytes = (size_t)n * (size_t)lay.stride;
    uint8_t *raw, *rgb;
    double *x, *y, *z;
    float* inten = nullptr;
    if ((rc = sc.get(&raw, raw_bytes)) || (rc = sc.get(&x, n)) || (rc = sc.get(&y, n)) || (rc = sc.get(&z, n)) || (rc = sc.get(&rgb, 3 * n)) ||
        (with_intensity && (rc = sc.get(&inten, n))))
      return rc;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return ctx->fail(PCV_E_IO, "Could not open input file.");
    const off_t body = (off_t)lay.body_offset;
    rc = ctx->h2d_fill(raw, raw_bytes, [&](uint8_t* to, size_t off, size_t len) {
      size_t got = 0;
      while (got < len) {  // pread is thread-safe: the host threads fill different parts of the chunk
        const ssize_t r = pread(fd, to + got, len - got, body + (off_t)(off + got));
        if (r <= 0) return false;
        got += (size_t)r;
      }
      return true;
    });
    close(fd);
    if (rc) return rc == PCV_E_IO ? ctx->fail(PCV_E_IO, "unexpected end of file in the vertex data") : rc;
    PlyDecodeArgs a{lay.stride, lay.x_type, lay.x_off, lay.y_type, lay.y_off, lay.z_type, lay.z_off, lay.r_off, lay.g_off, lay.b_off,
                    lay.i_off, {lay.offset[0], lay.offset[1], lay.offset[2]}};
    hipLaunchKernelGGL(ply_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, n, raw, x, y, z, rgb, inten);
    PCV_HIP_CHECK(ctx, hipGetLastError());
    // the raw vertex records are dead once the decode kernel is queued: the pool is stream-ordered, so the build below may
    // recycle the block (stride x n bytes: 15 B/pt of a float xyz + uchar rgb file) instead of holding it next to the
    // decoded arrays for its whole run
    sc.detach(raw);
    ctx->dev_free(raw);
    pts.x = x, pts.y = y, pts.z = z;
    pts.color = rgb;
    pts.intensity = inten;
  }
  pcv_build_params p = *params;
  p.flags |= PCV_BUILD_COMPUTE_BBOX;  // find_bounding_box over the file's points (generation.rs:256-270)
  return pcv_build_octree(ctx, &p, &pts, out);  // synchronous: the decoded arrays (scratch) outlive every kernel that reads them
}
