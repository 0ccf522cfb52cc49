// pcv_io.cpp — writes a finished octree in the reference's on-disk layout (host side, no GPU work).
//
//   <NodeId>.xyz        n * 3 * {1,2,4,8} bytes, little endian, AoS xyz   reference src/read_write/raw.rs:374-392
//   <NodeId>.rgb        n * 3 bytes                                        src/lib.rs:74-80 (attribute_extension)
//   <NodeId>.intensity  n * 4 bytes LE f32
//   files of a node with zero points do not exist                          src/read_write/node_writer.rs:78-89
//   meta.pb             proto3 `Meta`, version 13                          point_viewer_proto_rust/src/proto.proto:136-149
// NodeId Display: "r" + index in octal, zero-padded to `level` digits      src/octree/node.rs:73-86
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pcv_internal.h"

namespace {

std::string node_name(const pcv_node_info& n) {
  std::string s = "r";
  unsigned __int128 index = ((unsigned __int128)(n.id_high & 0x00ffffffffffffffull) << 64) | n.id_low;
  for (int j = (int)n.level - 1; j >= 0; --j) s.push_back((char)('0' + (int)((index >> (3 * j)) & 7)));
  return s;
}

void varint(std::vector<uint8_t>& o, uint64_t v) {
  while (v >= 0x80) {
    o.push_back((uint8_t)(v | 0x80));
    v >>= 7;
  }
  o.push_back((uint8_t)v);
}
void tag(std::vector<uint8_t>& o, int field, int wire) { varint(o, ((uint64_t)field << 3) | (uint64_t)wire); }
void f64_field(std::vector<uint8_t>& o, int field, double d) {
  if (d == 0.) return;  // proto3: default values are not serialised
  tag(o, field, 1);
  uint64_t u;
  std::memcpy(&u, &d, 8);
  for (int i = 0; i < 8; ++i) o.push_back((uint8_t)(u >> (8 * i)));
}
void bytes_field(std::vector<uint8_t>& o, int field, const std::vector<uint8_t>& b) {
  tag(o, field, 2);
  varint(o, b.size());
  o.insert(o.end(), b.begin(), b.end());
}
std::vector<uint8_t> vec3d(const double v[3]) {  // proto.proto:33-37 Vector3d
  std::vector<uint8_t> o;
  f64_field(o, 1, v[0]);
  f64_field(o, 2, v[1]);
  f64_field(o, 3, v[2]);
  return o;
}

bool write_file(const std::string& path, const uint8_t* data, uint64_t len) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  bool ok = len == 0 || fwrite(data, 1, len, f) == len;
  return fclose(f) == 0 && ok;
}
// one file of a node: open relative to the directory handle (no path walk), one write, close
bool write_file_at(int dirfd, const std::string& name, const uint8_t* data, uint64_t len) {
  const int fd = openat(dirfd, name.c_str(), O_CREAT | O_WRONLY | O_TRUNC | O_CLOEXEC, 0666);
  if (fd < 0) return false;
  bool ok = true;
  while (len) {
    const ssize_t w = ::write(fd, data, (size_t)len);
    if (w <= 0) {
      ok = false;
      break;
    }
    data += w;
    len -= (uint64_t)w;
  }
  return ::close(fd) == 0 && ok;
}

}  // namespace

// octree/mod.rs:87-99 to_meta_proto + node.rs:260-270 to_node_proto + node.rs:101-106 NodeId::to_proto
static std::vector<uint8_t> encode_meta(double resolution, const double bbox_min[3], const double bbox_max[3],
                                        const pcv_node_info* nodes, size_t count) {
  std::vector<uint8_t> octree;
  f64_field(octree, 2, resolution);
  for (size_t k = 0; k < count; ++k) {
    const pcv_node_info& n = nodes[k];
    std::vector<uint8_t> node, id;
    tag(node, 2, 0);
    varint(node, n.encoding);
    if (n.num_points != 0) {
      tag(node, 3, 0);
      varint(node, (uint64_t)n.num_points);
    }
    if (n.id_high != 0) {
      tag(id, 3, 0);
      varint(id, n.id_high);
    }
    if (n.id_low != 0) {
      tag(id, 4, 0);
      varint(id, n.id_low);
    }
    bytes_field(node, 4, id);  // the id sub-message is always present (octree/mod.rs:199 unwraps it)
    bytes_field(octree, 3, node);
  }
  std::vector<uint8_t> cuboid;
  bytes_field(cuboid, 3, vec3d(bbox_min));
  bytes_field(cuboid, 4, vec3d(bbox_max));
  std::vector<uint8_t> meta;
  tag(meta, 1, 0);
  varint(meta, 13);  // CURRENT_VERSION src/lib.rs:48
  bytes_field(meta, 4, cuboid);
  bytes_field(meta, 6, octree);
  return meta;
}
std::vector<uint8_t> pcv_encode_meta(const pcv_octree* t) {
  return encode_meta(t->resolution, t->bbox_min, t->bbox_max, t->nodes.data(), t->nodes.size());
}

extern "C" int pcv_write_meta(const char* directory, double resolution, const double bbox_min[3], const double bbox_max[3],
                              const pcv_node_info* nodes, uint64_t count) {
  if (!directory || !bbox_min || !bbox_max || (count && !nodes)) return PCV_E_INVALID;
  std::string dir(directory);
  ::mkdir(dir.c_str(), 0777);
  std::vector<uint8_t> meta = encode_meta(resolution, bbox_min, bbox_max, nodes, (size_t)count);
  return write_file(dir + "/meta.pb", meta.data(), meta.size()) ? PCV_OK : PCV_E_IO;
}

static int write_nodes(pcv_octree* t, const char* directory, uint32_t min_level, bool with_meta);
extern "C" int pcv_octree_write_dir(pcv_octree* t, const char* directory) { return write_nodes(t, directory, 0, true); }
extern "C" int pcv_octree_write_nodes(pcv_octree* t, const char* directory, uint32_t min_level) {
  return write_nodes(t, directory, min_level, false);
}
static int write_nodes(pcv_octree* t, const char* directory, uint32_t min_level, bool with_meta) {
  if (!t || !directory) return PCV_E_INVALID;
  pcv_ctx* ctx = t->ctx;
  std::string dir(directory);
  ::mkdir(dir.c_str(), 0777);  // generation.rs:308 "Ignore errors, maybe directory is already there."
  struct stat st;
  if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return ctx->fail(PCV_E_IO, "cannot create directory " + dir);
  const int dirfd = open(dir.c_str(), O_RDONLY | O_DIRECTORY | O_CLOEXEC);
  if (dirfd < 0) return ctx->fail(PCV_E_IO, "cannot open directory " + dir);
  // thousands of small files — written by a pool of host threads while the node blobs are still coming
  // down: the D2H is queued in chunks with an event each, a writer only waits for the chunks its node ends in (the
  // blobs are node-contiguous and in table order, so the copy runs ahead of the writers).
  constexpr uint64_t kChunk = 16ull << 20;
  struct Blob {
    const uint8_t* dev;
    uint8_t* host;
    uint64_t bytes;
    std::vector<hipEvent_t> ev;
  } blobs[3] = {{t->d_xyz, nullptr, t->xyz_bytes, {}}, {t->d_rgb, nullptr, t->rgb_bytes, {}}, {t->d_int, nullptr, t->int_bytes, {}}};
  int rc = PCV_OK;
  const bool stream_down = !t->host_valid && t->directory.empty();
  if (stream_down) {
    if (hipSetDevice(ctx->device) != hipSuccess) rc = ctx->fail(PCV_E_HIP, "hipSetDevice");
    if (!rc && t->xyz_bytes) rc = ctx->host_alloc((void**)&t->h_xyz.p, t->xyz_bytes);
    if (!rc && t->rgb_bytes) rc = ctx->host_alloc((void**)&t->h_rgb.p, t->rgb_bytes);
    if (!rc && t->int_bytes) rc = ctx->host_alloc((void**)&t->h_int.p, t->int_bytes);
    blobs[0].host = t->h_xyz.p, blobs[1].host = t->h_rgb.p, blobs[2].host = t->h_int.p;
    // interleave the chunks of the three blobs so that all of a node's files become writable at about the same time
    const uint64_t most = std::max(std::max(blobs[0].bytes, blobs[1].bytes), blobs[2].bytes);
    for (uint64_t off = 0; !rc && off < most; off += kChunk)
      for (Blob& b : blobs) {
        const uint64_t boff = b.bytes * (off / kChunk) / ((most + kChunk - 1) / kChunk);  // proportional progress
        const uint64_t bend = b.bytes * (off / kChunk + 1) / ((most + kChunk - 1) / kChunk);
        hipEvent_t e = nullptr;  // one event per chunk index and blob, also where the blob has nothing in this chunk
        if ((bend > boff && hipMemcpyAsync(b.host + boff, b.dev + boff, bend - boff, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) ||
            hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, ctx->stream) != hipSuccess) {
          if (e) (void)hipEventDestroy(e);  // created but never pushed
          rc = ctx->fail(PCV_E_HIP, "queueing the blob download failed");
          break;
        }
        b.ev.push_back(e);
      }
  } else if (!t->host_valid) {
    rc = ctx->fail(PCV_E_INVALID, "this octree was opened from a directory and its node files were not loaded; there is nothing to write");
  }
  // chunk k of a blob covers bytes [bytes * k / chunks, bytes * (k + 1) / chunks). The calling thread waits for the
  // events in order and publishes how many chunks have arrived; the writers only read that counter (no HIP call in a
  // writer thread: the first HIP call of a new thread costs milliseconds of runtime set-up).
  const uint64_t chunks = std::max<uint64_t>(1, blobs[0].ev.size());
  std::atomic<uint64_t> arrived{stream_down ? 0 : chunks};
  std::atomic<int> download_failed{0};
  auto wait_for = [&](int which, uint64_t end_byte) -> bool {  // everything before end_byte of blob `which` is on the host
    if (!stream_down || end_byte == 0) return true;
    const Blob& b = blobs[which];
    uint64_t need = 1;  // chunks that must have arrived
    while (need < chunks && b.bytes * need / chunks < end_byte) ++need;
    for (int spin = 0; arrived.load(std::memory_order_acquire) < need; ++spin) {
      if (download_failed.load()) return false;
      if (spin > 64) std::this_thread::sleep_for(std::chrono::microseconds(20));
      else std::this_thread::yield();
    }
    return true;
  };
  const size_t count = t->nodes.size();
  unsigned nthreads = std::thread::hardware_concurrency();
  if (nthreads == 0) nthreads = 4;
  // creating files in ONE directory serialises on the directory lock: past ~8 writers the lock convoy costs more than
  // the extra copies gain (measured on tmpfs: 8 threads 40 ms, 32 threads 50 ms, 64 threads 144 ms for 12 009 files)
  if (nthreads > 8) nthreads = 8;
  if (const char* e = pcv_experiment("PCV_WRITER_THREADS")) nthreads = (unsigned)std::max(1, atoi(e));  // experiments
  if (nthreads > count) nthreads = count ? (unsigned)count : 1;
  std::atomic<size_t> next{0};
  std::atomic<int> failed{0};
  std::string first_error;
  std::mutex err_mu;
  auto worker = [&]() {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= count || failed.load()) return;
      const pcv_node_info& n = t->nodes[i];
      if (n.num_points == 0 || n.level < min_level) continue;  // node_writer.rs:78-89: empty nodes have no files
      const std::string stem = node_name(n);
      const uint64_t np = (uint64_t)n.num_points;
      const uint64_t xb = np * 3 * (uint64_t)pcv_bytes_per_coordinate(n.encoding);
      const char* bad = nullptr;
      if (!wait_for(0, n.xyz_offset + xb) || !wait_for(1, (n.point_offset + np) * 3) ||
          (t->has_intensity && !wait_for(2, (n.point_offset + np) * 4)))
        bad = " (download)";
      else if (!write_file_at(dirfd, stem + ".xyz", t->h_xyz.data() + n.xyz_offset, xb))
        bad = ".xyz";
      else if (!write_file_at(dirfd, stem + ".rgb", t->h_rgb.data() + n.point_offset * 3, np * 3))
        bad = ".rgb";
      else if (t->has_intensity && !write_file_at(dirfd, stem + ".intensity", t->h_int.data() + n.point_offset * 4, np * 4))
        bad = ".intensity";
      if (bad) {
        std::lock_guard<std::mutex> g(err_mu);
        if (!failed.exchange(1)) first_error = "cannot write " + dir + "/" + stem + bad;
      }
    }
  };
  if (!rc) {
    std::vector<std::thread> pool;
    for (unsigned k = stream_down ? 0 : 1; k < nthreads; ++k) pool.emplace_back(worker);
    if (stream_down) {  // this thread follows the download and publishes its progress
      for (uint64_t k = 0; k < chunks; ++k) {
        bool ok = true;
        for (Blob& b : blobs)
          if (k < b.ev.size() && hipEventSynchronize(b.ev[k]) != hipSuccess) ok = false;
        if (!ok) {
          download_failed.store(1);
          break;
        }
        arrived.store(k + 1, std::memory_order_release);
      }
    } else {
      worker();
    }
    for (auto& th : pool) th.join();
    if (download_failed.load()) rc = ctx->fail(PCV_E_HIP, "blob download failed");
  }
  if (stream_down) {
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = ctx->fail(PCV_E_HIP, "blob download failed");
    for (Blob& b : blobs)
      for (hipEvent_t e : b.ev) (void)hipEventDestroy(e);
    if (!rc) {
      t->host_valid = true;
    } else {  // a failed download leaves no half-filled host blobs attached to the tree (a retry would allocate over them)
      if (t->h_xyz.p) ctx->host_release(t->h_xyz.p);
      if (t->h_rgb.p) ctx->host_release(t->h_rgb.p);
      if (t->h_int.p) ctx->host_release(t->h_int.p);
      t->h_xyz.p = t->h_rgb.p = t->h_int.p = nullptr;
    }
  }
  ::close(dirfd);
  if (rc) return rc;
  if (failed.load()) return ctx->fail(PCV_E_IO, first_error);
  if (!with_meta) return PCV_OK;
  std::vector<uint8_t> meta = pcv_encode_meta(t);
  if (!write_file(dir + "/meta.pb", meta.data(), meta.size())) return ctx->fail(PCV_E_IO, "cannot write meta.pb");
  return PCV_OK;
}

// ------------------------------------------------------------------------------------------------
// Loading: Octree::from_data_provider over a directory (src/octree/mod.rs:156-215, data_provider/on_disk.rs)
// ------------------------------------------------------------------------------------------------
namespace {

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  uint64_t varint() {
    uint64_t v = 0;
    for (int s = 0; p < end && s < 64; s += 7) {
      uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << s;
      if (!(b & 0x80)) return v;
    }
    ok = false;
    return 0;
  }
  double f64() {
    if (end - p < 8) {
      ok = false;
      return 0;
    }
    uint64_t u = 0;
    for (int i = 0; i < 8; ++i) u |= (uint64_t)p[i] << (8 * i);
    p += 8;
    double d;
    std::memcpy(&d, &u, 8);
    return d;
  }
  Reader sub() {
    uint64_t len = varint();
    if ((uint64_t)(end - p) < len) {
      ok = false;
      len = 0;
    }
    Reader r{p, p + len};
    p += len;
    return r;
  }
  void skip(int wire) {
    if (wire == 0) varint();
    else if (wire == 1 && end - p >= 8) p += 8;
    else if (wire == 2) sub();
    else if (wire == 5 && end - p >= 4) p += 4;
    else ok = false;
  }
};

// Vector3d (fixed64 fields) or Vector3f (fixed32 fields, proto.proto:27-31) into doubles
void read_vec3(Reader r, double v[3]) {
  v[0] = v[1] = v[2] = 0.;
  while (r.p < r.end && r.ok) {
    uint64_t t = r.varint();
    int f = (int)(t >> 3), w = (int)(t & 7);
    if (w == 1 && f >= 1 && f <= 3) v[f - 1] = r.f64();
    else if (w == 5 && f >= 1 && f <= 3 && r.end - r.p >= 4) {
      uint32_t u = 0;
      for (int i = 0; i < 4; ++i) u |= (uint32_t)r.p[i] << (8 * i);
      r.p += 4;
      float fl;
      std::memcpy(&fl, &u, 4);
      v[f - 1] = (double)fl;  // proto::Vector3d::from(Vector3f): plain widening
    } else r.skip(w);
  }
}
// impl From<&proto::AxisAlignedCuboid> for Aabb (src/geometry/aabb.rs:69-84): min/max (fields 3/4, Vector3d), else the
// deprecated Vector3f fields 1/2 of version <= 10
void read_cuboid(Reader c, double mn[3], double mx[3]) {
  bool have_min = false, have_max = false;
  double dmn[3] = {0, 0, 0}, dmx[3] = {0, 0, 0};
  while (c.p < c.end && c.ok) {
    uint64_t t = c.varint();
    int f = (int)(t >> 3), w = (int)(t & 7);
    if (f == 3 && w == 2) {
      read_vec3(c.sub(), mn);
      have_min = true;
    } else if (f == 4 && w == 2) {
      read_vec3(c.sub(), mx);
      have_max = true;
    } else if (f == 1 && w == 2) read_vec3(c.sub(), dmn);
    else if (f == 2 && w == 2) read_vec3(c.sub(), dmx);
    else c.skip(w);
  }
  for (int a = 0; a < 3; ++a) {
    if (!have_min) mn[a] = dmn[a];
    if (!have_max) mx[a] = dmx[a];
  }
}

bool read_file(const std::string& path, std::vector<uint8_t>* out, bool* missing) {
  *missing = false;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) {
    *missing = true;
    return false;
  }
  long sz = -1;
  if (fseek(f, 0, SEEK_END) == 0) sz = ftell(f);
  if (sz < 0 || fseek(f, 0, SEEK_SET) != 0) {  // not seekable (a directory, a pipe): an I/O error, not a huge resize
    fclose(f);
    return false;
  }
  size_t got = 0;
  try {
    out->resize((size_t)sz);
    got = sz ? fread(out->data(), 1, (size_t)sz, f) : 0;
  } catch (...) {  // std::bad_alloc must not cross the C ABI
    fclose(f);
    return false;
  }
  fclose(f);
  return got == (size_t)sz;
}

}  // namespace

extern "C" int pcv_octree_open_dir(pcv_ctx* ctx, const char* directory, pcv_octree** out) {
  if (!ctx) return PCV_E_INVALID;
  if (!directory || !out) return ctx->fail(PCV_E_INVALID, "null argument");
  *out = nullptr;
  std::string dir(directory);
  std::vector<uint8_t> buf;
  bool missing;
  if (!read_file(dir + "/meta.pb", &buf, &missing))
    return ctx->fail(missing ? PCV_E_NOT_FOUND : PCV_E_IO, "cannot read " + dir + "/meta.pb");
  int version = 0;
  double bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0}, resolution = 0;
  double obmin[3] = {0, 0, 0}, obmax[3] = {0, 0, 0}, old_resolution = 0;  // version 12 / versions <= 11
  bool has_octree = false;
  struct RawNode {
    uint64_t hi, lo;
    int64_t num_points;
    uint32_t enc;
  };
  std::vector<RawNode> raw, old_raw;
  // proto::OctreeNode (proto.proto:90-94) incl. the NodeId of version 9 (deprecated_level / deprecated_index,
  // NodeId::from_proto node.rs:89-99)
  auto read_node = [](Reader n, RawNode* rn) {
    *rn = RawNode{0, 0, 0, 0};
    while (n.p < n.end && n.ok) {
      uint64_t nt = n.varint();
      int nf = (int)(nt >> 3), nw = (int)(nt & 7);
      if (nf == 2 && nw == 0) rn->enc = (uint32_t)n.varint();
      else if (nf == 3 && nw == 0) rn->num_points = (int64_t)n.varint();
      else if (nf == 4 && nw == 2) {
        Reader id = n.sub();
        uint64_t dep_level = 0, dep_index = 0;
        while (id.p < id.end && id.ok) {
          uint64_t it = id.varint();
          int idf = (int)(it >> 3), idw = (int)(it & 7);
          if (idf == 3 && idw == 0) rn->hi = id.varint();
          else if (idf == 4 && idw == 0) rn->lo = id.varint();
          else if (idf == 1 && idw == 0) dep_level = id.varint() & 0xffu;  // `as u8`
          else if (idf == 2 && idw == 0) dep_index = id.varint();
          else id.skip(idw);
        }
        if (!id.ok) n.ok = false;
        if (dep_level != 0 || dep_index != 0) {  // from_level_index(level, index as u128): i64 -> u128 sign-extends
          rn->hi = (dep_level << 56) | (((int64_t)dep_index < 0) ? ~0ull : 0ull);
          rn->lo = dep_index;
        }
      } else n.skip(nw);
    }
    return n.ok;
  };
  Reader r{buf.data(), buf.data() + buf.size()};
  while (r.p < r.end && r.ok) {
    uint64_t t = r.varint();
    int f = (int)(t >> 3), w = (int)(t & 7);
    if (f == 1 && w == 0) version = (int)r.varint();
    else if (f == 4 && w == 2) read_cuboid(r.sub(), bmin, bmax);
    else if (f == 3 && w == 1) old_resolution = r.f64();  // deprecated_resolution (versions <= 11)
    else if (f == 5 && w == 2) {                          // deprecated_nodes (versions <= 11)
      RawNode rn;
      if (!read_node(r.sub(), &rn)) r.ok = false;
      old_raw.push_back(rn);
    } else if (f == 6 && w == 2) {
      has_octree = true;
      Reader o = r.sub();
      while (o.p < o.end && o.ok) {
        uint64_t ot = o.varint();
        int of = (int)(ot >> 3), ow = (int)(ot & 7);
        if (of == 2 && ow == 1) resolution = o.f64();
        else if (of == 1 && ow == 2) read_cuboid(o.sub(), obmin, obmax);  // deprecated_bounding_box (version 12)
        else if (of == 3 && ow == 2) {
          RawNode rn;
          if (!read_node(o.sub(), &rn)) o.ok = false;
          raw.push_back(rn);
        } else o.skip(ow);
      }
      if (!o.ok) r.ok = false;
    } else r.skip(w);
  }
  if (!r.ok) return ctx->fail(PCV_E_INVALID, "Could not parse meta.pb");
  // Octree::from_data_provider (octree/mod.rs:156-215): 9 | 10 | 11 read the top-level fields, 12 | 13 the OctreeMeta
  if (version >= 9 && version <= 11) {
    resolution = old_resolution;
    raw.swap(old_raw);
  } else if (version == 12 || version == 13) {
    if (!has_octree) return ctx->fail(PCV_E_INVALID, "No octree meta found");
    if (version == 12)
      for (int a = 0; a < 3; ++a) {
        bmin[a] = obmin[a];
        bmax[a] = obmax[a];
      }
  } else {
    return ctx->fail(PCV_E_INVALID, "InvalidVersion(" + std::to_string(version) + ")");
  }
  {
    // meta.pb is untrusted input: the offsets derived from num_points become raw write targets when the node files are
    // loaded (pcv_octree_load_device), so a negative count or a sum that wraps must never get that far. 2^56 bytes of
    // blob is far beyond any device and keeps every product below (x 3 coordinates x 8 bytes + padding) inside 64 bits.
    constexpr uint64_t kMaxBlob = 1ull << 56;
    uint64_t total_bytes = 0;
    for (const RawNode& rn : raw) {
      if ((rn.hi >> 56) > (uint64_t)PCV_MAX_LEVELS)  // 120 index bits name 40 levels (node.rs:101-111); also keeps every shift below in range
        return ctx->fail(PCV_E_INVALID, "meta.pb: node level " + std::to_string(rn.hi >> 56) + " exceeds what a NodeId can name");
      if (rn.num_points < 0) return ctx->fail(PCV_E_INVALID, "meta.pb: negative num_points");
      if ((uint64_t)rn.num_points > kMaxBlob / 32) return ctx->fail(PCV_E_INVALID, "meta.pb: num_points out of range");
      total_bytes += (uint64_t)rn.num_points * 24 + 16;  // widest encoding + the 16-byte padding of a node's xyz block
      if (total_bytes > kMaxBlob) return ctx->fail(PCV_E_INVALID, "meta.pb: the nodes' num_points add up to more than 2^56 bytes");
    }
  }

  pcv_octree* t = new pcv_octree();
  t->ctx = ctx;
  t->resolution = resolution;
  t->directory = dir;
  for (int a = 0; a < 3; ++a) {  // Aabb::new: inf / sup
    t->bbox_min[a] = std::fmin(bmin[a], bmax[a]);
    t->bbox_max[a] = std::fmax(bmin[a], bmax[a]);
  }
  const double root_edge = std::fmax(std::fmax(t->bbox_max[0] - t->bbox_min[0], t->bbox_max[1] - t->bbox_min[1]),
                                     t->bbox_max[2] - t->bbox_min[2]);
  typedef unsigned __int128 u128;
  for (const RawNode& rn : raw) {
    if (rn.enc < 1 || rn.enc > 4) {  // codec.rs:50-53 PositionEncoding::from_proto(INVALID)
      delete t;
      return ctx->fail(PCV_E_INVALID, "Proto: PositionEncoding is invalid");
    }
    pcv_node_info ni{};
    ni.id_high = rn.hi;
    ni.id_low = rn.lo;
    ni.num_points = rn.num_points;
    ni.level = (uint32_t)(rn.hi >> 56);
    ni.encoding = rn.enc;
    // NodeId::find_bounding_cube (node.rs:157-172)
    u128 v = ((u128)rn.hi << 64) | rn.lo;
    double edge = root_edge, mn[3] = {t->bbox_min[0], t->bbox_min[1], t->bbox_min[2]};
    for (int level = (int)ni.level - 1; level >= 0; --level) {
      edge /= 2.;
      unsigned ci = (unsigned)((v >> (3 * level)) & 7);
      mn[0] += (double)((ci >> 2) & 1) * edge;
      mn[1] += (double)((ci >> 1) & 1) * edge;
      mn[2] += (double)(ci & 1) * edge;
    }
    for (int a = 0; a < 3; ++a) ni.cube_min[a] = mn[a];
    ni.cube_edge = edge;
    t->nodes.push_back(ni);
    t->num_points += (uint64_t)rn.num_points;
  }
  std::sort(t->nodes.begin(), t->nodes.end(), [](const pcv_node_info& a, const pcv_node_info& b) {
    if (a.level != b.level) return a.level < b.level;
    const uint64_t ah = a.id_high & 0x00ffffffffffffffull, bh = b.id_high & 0x00ffffffffffffffull;
    if (ah != bh) return ah < bh;
    return a.id_low < b.id_low;
  });
  uint64_t point_off = 0, xyz_off = 0;
  for (pcv_node_info& ni : t->nodes) {
    ni.point_offset = point_off;
    ni.xyz_offset = xyz_off;
    point_off += (uint64_t)ni.num_points;
    xyz_off += ((uint64_t)ni.num_points * 3 * (uint64_t)pcv_bytes_per_coordinate(ni.encoding) + 15) & ~15ull;  // as built trees
  }
  // intensity is implied by the presence of the root's .intensity file (octree/mod.rs:57-74 hard-codes both)
  struct stat st;
  t->has_intensity = stat((dir + "/r.intensity").c_str(), &st) == 0;
  *out = t;
  return PCV_OK;
}

// Octree::get_node_data (octree/mod.rs:285-307): raw file content; missing file -> NodeNotFound.
int pcv_octree_read_node_file(pcv_octree* t, uint64_t i, int which, const uint8_t** data, uint64_t* len) {
  pcv_ctx* ctx = t->ctx;
  auto key = std::make_pair(i, which);
  auto it = t->file_cache.find(key);
  if (it == t->file_cache.end()) {
    const char* ext = which == 0 ? ".xyz" : (which == 1 ? ".rgb" : ".intensity");
    std::vector<uint8_t> buf;
    bool missing;
    const std::string path = t->directory + "/" + node_name(t->nodes[i]) + ext;
    if (!read_file(path, &buf, &missing)) {
      if (missing && (t->nodes[i].num_points == 0 || which == 2)) buf.clear();  // empty nodes have no files
      else return ctx->fail(missing ? PCV_E_NOT_FOUND : PCV_E_IO, "cannot read " + path);
    }
    it = t->file_cache.emplace(key, std::move(buf)).first;
  }
  *data = it->second.data();
  *len = it->second.size();
  return PCV_OK;
}

// N3 on an octree that lives on disk (src/iterator.rs:185-223 -> Octree::points_in_node, octree/mod.rs:285-307 ->
// NodeIterator, read_write/node_iterator.rs:24-119): every node file is read once (pool of host threads, straight
// into pinned blobs laid out like a built tree's) and uploaded; the cull / query kernels then decode on load exactly
// as they do for a tree built in this process. File sizes are checked against meta.pb's num_points.
int pcv_octree_load_device(pcv_octree* t) {
  pcv_ctx* ctx = t->ctx;
  if (t->d_xyz || t->directory.empty() || t->nodes.empty()) return PCV_OK;
  const pcv_node_info& last = t->nodes.back();
  t->xyz_bytes = last.xyz_offset + (((uint64_t)last.num_points * 3 * (uint64_t)pcv_bytes_per_coordinate(last.encoding) + 15) & ~15ull);
  t->rgb_bytes = t->num_points * 3;
  t->int_bytes = t->has_intensity ? t->num_points * 4 : 0;
  if (t->num_points == 0) return PCV_OK;
  PCV_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc;
  auto drop_host = [&]() {  // a failed attempt leaves nothing behind: the next query starts from scratch
    if (t->h_xyz.p) ctx->host_release(t->h_xyz.p);
    if (t->h_rgb.p) ctx->host_release(t->h_rgb.p);
    if (t->h_int.p) ctx->host_release(t->h_int.p);
    t->h_xyz.p = t->h_rgb.p = t->h_int.p = nullptr;
  };
  if ((rc = ctx->host_alloc((void**)&t->h_xyz.p, t->xyz_bytes)) || (rc = ctx->host_alloc((void**)&t->h_rgb.p, t->rgb_bytes)) ||
      (t->int_bytes && (rc = ctx->host_alloc((void**)&t->h_int.p, t->int_bytes)))) {
    drop_host();
    return rc;
  }
  const size_t count = t->nodes.size();
  unsigned nthreads = std::thread::hardware_concurrency();
  if (nthreads == 0) nthreads = 4;
  if (nthreads > 32) nthreads = 32;
  if (nthreads > count) nthreads = (unsigned)count;
  std::atomic<size_t> next{0};
  std::atomic<int> failed{0};
  std::string first_error;
  std::mutex err_mu;
  auto read_into = [](const std::string& path, uint8_t* dst, uint64_t want) -> const char* {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return "cannot open ";
    const size_t got = fread(dst, 1, (size_t)want, f);
    uint8_t extra;
    const bool longer = got == want && fread(&extra, 1, 1, f) == 1;
    fclose(f);
    if (got != want || longer) return "size does not match meta.pb's num_points: ";
    return nullptr;
  };
  auto worker = [&]() {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= count || failed.load()) return;
      const pcv_node_info& n = t->nodes[i];
      if (n.num_points <= 0) continue;  // node_writer.rs:78-89: empty nodes have no files
      const std::string stem = t->directory + "/" + node_name(n);
      const uint64_t np = (uint64_t)n.num_points;
      const uint64_t xb = np * 3 * (uint64_t)pcv_bytes_per_coordinate(n.encoding);
      const char* bad = read_into(stem + ".xyz", t->h_xyz.p + n.xyz_offset, xb);
      std::string which = ".xyz";
      if (!bad) {
        std::memset(t->h_xyz.p + n.xyz_offset + xb, 0, (size_t)(((xb + 15) & ~15ull) - xb));
        bad = read_into(stem + ".rgb", t->h_rgb.p + n.point_offset * 3, np * 3);
        which = ".rgb";
      }
      if (!bad && t->has_intensity) {
        bad = read_into(stem + ".intensity", t->h_int.p + n.point_offset * 4, np * 4);
        which = ".intensity";
      }
      if (bad) {
        std::lock_guard<std::mutex> g(err_mu);
        if (!failed.exchange(1)) first_error = std::string(bad) + stem + which;
      }
    }
  };
  std::vector<std::thread> pool;
  for (unsigned k = 1; k < nthreads; ++k) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  if (failed.load()) {
    drop_host();
    return ctx->fail(PCV_E_IO, first_error);
  }
  // device blobs: into locals first — t->d_xyz is what "already loaded" is read off, so it is only set once every
  // allocation and upload has succeeded
  void *dx = nullptr, *dr = nullptr, *di = nullptr;
  auto fail_dev = [&](int code, const char* what) {
    if (dx) ctx->dev_free(dx);
    if (dr) ctx->dev_free(dr);
    if (di) ctx->dev_free(di);
    drop_host();
    return what ? ctx->fail(code, what) : code;
  };
  if ((rc = ctx->dev_alloc(&dx, t->xyz_bytes)) || (rc = ctx->dev_alloc(&dr, t->rgb_bytes)) ||
      (t->int_bytes && (rc = ctx->dev_alloc(&di, t->int_bytes))))
    return fail_dev(rc, nullptr);
  if (hipMemcpyAsync(dx, t->h_xyz.p, t->xyz_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipMemcpyAsync(dr, t->h_rgb.p, t->rgb_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      (t->int_bytes && hipMemcpyAsync(di, t->h_int.p, t->int_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    (void)hipStreamSynchronize(ctx->stream);  // nothing may still be reading the pinned blocks when they go back
    return fail_dev(PCV_E_HIP, "uploading the node files failed");
  }
  t->d_xyz = (uint8_t*)dx;
  t->d_rgb = (uint8_t*)dr;
  t->d_int = (uint8_t*)di;
  t->host_valid = true;
  return PCV_OK;
}
