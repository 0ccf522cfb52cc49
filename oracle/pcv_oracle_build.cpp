// pcv_oracle_build.cpp — CPU ORACLE for the octree build path (test infrastructure only).
//
// Two independent formulations of point_cloud_viewer's `build_octree`
// (src/octree/generation.rs:289-403), both producing the reference's on-disk layout:
//
//  * LITERAL  — streams batches through per-node files exactly like the reference: recursive 8-way
//               split with per-level quantise -> decode (generation.rs:58-193), bottom-up every-8th
//               promotion with child rewrite (generation.rs:195-253, 335-387), meta.pb
//               (generation.rs:390-402). Backend = real directory (for CPU-baseline timing and
//               directory parity) or an in-memory file map (for fast parity).
//  * CLOSED   — the formulation the HIP pipeline implements: topology-independent per-point path
//               digits ("chain keys"), stable per-node lists, closed-form promotion slots, byte replay
//               of the decode/encode chain.
//
// tests/test_oracle_modes.py keeps the two in lock-step (nodes, counts, bytes).
// Nothing under point_cloud_viewer_amd/ may link or call this file.
#include <omp.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <unordered_map>

#include "pcv_oracle_core.h"

namespace pcvo {

int64_t MAX_POINTS_PER_NODE = 100000;

// ------------------------------------------------------------------------------------------------
// Storage backend: a directory on disk or an in-memory file map.
// ------------------------------------------------------------------------------------------------
struct Backend {
  bool disk = false;
  std::string dir;
  std::mutex mu;
  std::unordered_map<std::string, std::shared_ptr<std::vector<uint8_t>>> mem;

  std::string path(const std::string& name) const { return dir + "/" + name; }

  bool file_size(const std::string& name, uint64_t* sz) {
    if (disk) {
      struct stat st;
      if (stat(path(name).c_str(), &st) != 0) return false;
      *sz = (uint64_t)st.st_size;
      return true;
    }
    std::lock_guard<std::mutex> g(mu);
    auto it = mem.find(name);
    if (it == mem.end()) return false;
    *sz = it->second->size();
    return true;
  }
  void remove(const std::string& name) {
    if (disk) {
      ::unlink(path(name).c_str());
      return;
    }
    std::lock_guard<std::mutex> g(mu);
    mem.erase(name);
  }
  bool read_all(const std::string& name, std::vector<uint8_t>* out) {
    if (disk) {
      FILE* f = fopen(path(name).c_str(), "rb");
      if (!f) return false;
      fseek(f, 0, SEEK_END);
      long sz = ftell(f);
      fseek(f, 0, SEEK_SET);
      out->resize((size_t)sz);
      size_t got = sz ? fread(out->data(), 1, (size_t)sz, f) : 0;
      fclose(f);
      return got == (size_t)sz;
    }
    std::lock_guard<std::mutex> g(mu);
    auto it = mem.find(name);
    if (it == mem.end()) return false;
    *out = *it->second;
    return true;
  }
};

// node_writer.rs:30-89 DataWriter: create+truncate on open, count bytes, delete the file on drop when
// nothing was written.
class DataWriter {
 public:
  DataWriter(Backend* be, const std::string& name) : be_(be), name_(name) {
    if (be_->disk) {
      f_ = fopen(be_->path(name).c_str(), "wb");
      if (!f_) {
        fprintf(stderr, "pcv oracle: cannot open %s\n", be_->path(name).c_str());
        abort();
      }
      setvbuf(f_, nullptr, _IOFBF, 8192);  // BufWriter default capacity
    } else {
      buf_ = std::make_shared<std::vector<uint8_t>>();
      std::lock_guard<std::mutex> g(be_->mu);
      be_->mem[name] = buf_;
    }
  }
  void write(const void* p, size_t n) {
    if (n == 0) return;
    if (f_) fwrite(p, 1, n, f_);
    else buf_->insert(buf_->end(), (const uint8_t*)p, (const uint8_t*)p + n);
    bytes_written_ += n;
  }
  uint64_t bytes_written() const { return bytes_written_; }
  ~DataWriter() {
    if (f_) fclose(f_);
    if (bytes_written_ == 0) be_->remove(name_);
  }

 private:
  Backend* be_;
  std::string name_;
  FILE* f_ = nullptr;
  std::shared_ptr<std::vector<uint8_t>> buf_;
  uint64_t bytes_written_ = 0;
};

// lib.rs:102-151 PointsBatch, restricted to the attributes the build path handles
// (color U8Vec3 always; intensity F32 optional — generation.rs / bin/build_octree.rs:47-52).
struct PointsBatch {
  std::vector<double> position;  // AoS xyz
  std::vector<uint8_t> color;    // AoS rgb
  std::vector<float> intensity;
  bool has_intensity = false;
  size_t len() const { return position.size() / 3; }
  // lib.rs:139-151 retain
  void retain_into(const std::vector<uint8_t>& keep, uint8_t want, PointsBatch* out) const {
    out->has_intensity = has_intensity;
    out->position.clear();
    out->color.clear();
    out->intensity.clear();
    size_t n = len();
    for (size_t i = 0; i < n; ++i) {
      if (keep[i] != want) continue;
      out->position.insert(out->position.end(), &position[3 * i], &position[3 * i] + 3);
      out->color.insert(out->color.end(), &color[3 * i], &color[3 * i] + 3);
      if (has_intensity) out->intensity.push_back(intensity[i]);
    }
  }
  void append(const PointsBatch& o) {  // lib.rs:109-126
    position.insert(position.end(), o.position.begin(), o.position.end());
    color.insert(color.end(), o.color.begin(), o.color.end());
    intensity.insert(intensity.end(), o.intensity.begin(), o.intensity.end());
  }
};

struct BuildCtx {
  Backend* be;
  double resolution;
  Aabb bbox;
  Cube root_cube;
  bool has_intensity;
  size_t batch_size;
};

// raw.rs:361-450 RawNodeWriter (+ generation.rs:39-56 from_data_provider): xyz writer opened
// (truncating) at construction, attribute writers lazily at first write, in BTreeMap key order
// ("color" < "intensity").
class RawNodeWriter {
 public:
  RawNodeWriter(const BuildCtx& ctx, NodeId id) : ctx_(ctx), stem_(id.to_string()) {
    cube_ = id.find_bounding_cube(ctx.root_cube);
    enc_ = position_encoding(cube_.edge, ctx.resolution);
    xyz_.reset(new DataWriter(ctx.be, stem_ + ".xyz"));
  }
  void write(const PointsBatch& b) {  // raw.rs:374-392 + node_writer.rs:281-316
    size_t n = b.len();
    int bpc = bytes_per_coordinate(enc_);
    scratch_.resize(n * 3 * (size_t)bpc);
    for (size_t i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a) {
        uint64_t raw = encode_coord(enc_, b.position[3 * i + a], cube_.mn[a], cube_.edge);
        put_le(&scratch_[(3 * i + a) * (size_t)bpc], raw, bpc);
      }
    xyz_->write(scratch_.data(), scratch_.size());
    if (!attrs_open_) {
      rgb_.reset(new DataWriter(ctx_.be, stem_ + ".rgb"));
      if (b.has_intensity) int_.reset(new DataWriter(ctx_.be, stem_ + ".intensity"));
      attrs_open_ = true;
    }
    rgb_->write(b.color.data(), b.color.size());
    if (b.has_intensity) int_->write(b.intensity.data(), b.intensity.size() * 4);
  }
  int64_t num_written() const {  // raw.rs:443-449
    return (int64_t)xyz_->bytes_written() / bytes_per_coordinate(enc_) / 3;
  }

 private:
  const BuildCtx& ctx_;
  std::string stem_;
  Cube cube_;
  Enc enc_;
  std::unique_ptr<DataWriter> xyz_, rgb_, int_;
  bool attrs_open_ = false;
  std::vector<uint8_t> scratch_;
};

// on_disk.rs:23-33 number_of_points: size of the .rgb file / 3, NodeNotFound when missing.
static bool number_of_points(const BuildCtx& ctx, NodeId id, int64_t* n) {
  uint64_t sz;
  if (!ctx.be->file_size(id.to_string() + ".rgb", &sz)) return false;
  *n = (int64_t)(sz / 3);
  return true;
}

// Stream of batches with a known total (lib.rs:55-57 NumberOfPoints).
struct Stream {
  virtual ~Stream() {}
  virtual size_t num_points() const = 0;
  virtual bool next(PointsBatch* b) = 0;
};

// The caller's input, chunked (stands in for `impl Iterator<Item = PointsBatch>` of generation.rs:293).
struct InputStream : Stream {
  const double *x, *y, *z;
  const uint8_t* rgb;
  const float* intensity;
  size_t n, pos = 0, batch;
  size_t num_points() const override { return n; }
  bool next(PointsBatch* b) override {
    if (pos >= n) return false;
    size_t m = std::min(batch, n - pos);
    b->has_intensity = intensity != nullptr;
    b->position.resize(3 * m);
    b->color.resize(3 * m);
    b->intensity.resize(intensity ? m : 0);
    for (size_t i = 0; i < m; ++i) {
      b->position[3 * i] = x[pos + i];
      b->position[3 * i + 1] = y[pos + i];
      b->position[3 * i + 2] = z[pos + i];
      b->color[3 * i] = rgb[3 * (pos + i)];
      b->color[3 * i + 1] = rgb[3 * (pos + i) + 1];
      b->color[3 * i + 2] = rgb[3 * (pos + i) + 2];
      if (intensity) b->intensity[i] = intensity[pos + i];
    }
    pos += m;
    return true;
  }
};

// node_iterator.rs:24-119 + raw.rs:127-344: read a node's files back in batches, decoding positions.
struct NodeIterator : Stream {
  std::vector<uint8_t> xyz, rgb, inten;
  Cube cube;
  Enc enc;
  size_t n, pos = 0, batch;
  bool has_intensity;
  NodeIterator(const BuildCtx& ctx, NodeId id, size_t num_points, size_t batch_size)
      : n(num_points), batch(batch_size), has_intensity(ctx.has_intensity) {
    cube = id.find_bounding_cube(ctx.root_cube);  // octree/mod.rs:76-84 encoding_for_node
    enc = position_encoding(cube.edge, ctx.resolution);
    std::string stem = id.to_string();
    if (n > 0) {
      if (!ctx.be->read_all(stem + ".xyz", &xyz) || !ctx.be->read_all(stem + ".rgb", &rgb)) {
        fprintf(stderr, "pcv oracle: node %s not found\n", stem.c_str());
        abort();
      }
      if (has_intensity && !ctx.be->read_all(stem + ".intensity", &inten)) {
        // generation.rs:167-177 .unwrap() — the reference panics here.
        fprintf(stderr, "pcv oracle: node %s has no intensity file\n", stem.c_str());
        abort();
      }
    }
  }
  size_t num_points() const override { return n; }
  bool next(PointsBatch* b) override {
    if (pos >= n) return false;
    size_t m = std::min(batch, n - pos);
    int bpc = bytes_per_coordinate(enc);
    b->has_intensity = has_intensity;
    b->position.resize(3 * m);
    b->color.assign(&rgb[3 * pos], &rgb[3 * pos] + 3 * m);
    b->intensity.resize(has_intensity ? m : 0);
    for (size_t i = 0; i < m; ++i)
      for (int a = 0; a < 3; ++a) {
        uint64_t raw = get_le(&xyz[(3 * (pos + i) + a) * (size_t)bpc], bpc);
        b->position[3 * i + a] = decode_coord(enc, raw, cube.mn[a], cube.edge);
      }
    if (has_intensity) std::memcpy(b->intensity.data(), &inten[4 * pos], 4 * m);
    pos += m;
    return true;
  }
};

// generation.rs:128-150
static bool should_split_node(const BuildCtx& ctx, NodeId id, int64_t num_points) {
  if (num_points <= MAX_POINTS_PER_NODE) return false;
  Cube c = id.find_bounding_cube(ctx.root_cube);
  if (c.edge <= ctx.resolution) return false;
  return true;
}

// generation.rs:58-126 split
static void split(const BuildCtx& ctx, NodeId node_id, Stream& stream, std::vector<NodeId>* leaf_nodes,
                  std::vector<NodeId>* split_nodes) {
  std::unique_ptr<RawNodeWriter> children[8];
  Cube bounding_cube = node_id.find_bounding_cube(ctx.root_cube);
  PointsBatch batch, child_batch;
  std::vector<uint8_t> child_indices;
  while (stream.next(&batch)) {
    size_t n = batch.len();
    child_indices.resize(n);
    for (size_t i = 0; i < n; ++i)
      child_indices[i] = child_index_from_bounding_cube(bounding_cube, &batch.position[3 * i]);
    for (uint8_t array_index = 0; array_index < 8; ++array_index) {
      batch.retain_into(child_indices, array_index, &child_batch);  // clone + retain, :85-90
      if (child_batch.len() != 0) {
        if (!children[array_index])
          children[array_index].reset(new RawNodeWriter(ctx, node_id.get_child_id(array_index)));
        children[array_index]->write(child_batch);
      }
    }
  }
  // generation.rs:104-108: reopen the node (truncating its .xyz) and drop it -> .xyz removed.
  { RawNodeWriter reopen(ctx, node_id); }
  for (uint8_t ci = 0; ci < 8; ++ci) {
    if (!children[ci]) continue;
    NodeId child_id = node_id.get_child_id(ci);
    if (should_split_node(ctx, child_id, children[ci]->num_written())) split_nodes->push_back(child_id);
    else leaf_nodes->push_back(child_id);
  }
}

struct LeafSink {
  std::mutex mu;
  std::vector<NodeId> leaves;
  void send(NodeId id) {
    std::lock_guard<std::mutex> g(mu);
    leaves.push_back(id);
  }
};

// generation.rs:152-193 split_node (rayon scope.spawn -> OpenMP task)
static void split_node(const BuildCtx& ctx, NodeId node_id, Stream& stream, LeafSink* sink) {
  std::vector<NodeId> leaf_nodes, split_nodes;
  split(ctx, node_id, stream, &leaf_nodes, &split_nodes);
  const BuildCtx* ctxp = &ctx;
  for (NodeId child_id : split_nodes) {
#pragma omp task firstprivate(child_id, ctxp, sink)
    {
      int64_t n = 0;
      number_of_points(*ctxp, child_id, &n);
      NodeIterator it(*ctxp, child_id, (size_t)n, NUM_POINTS_PER_BATCH);
      split_node(*ctxp, child_id, it, sink);
    }
  }
  for (NodeId id : leaf_nodes) sink->send(id);
}

struct FinishedSink {
  std::mutex mu;
  std::map<u128, int64_t> finished;
  void send(NodeId id, int64_t n) {
    std::lock_guard<std::mutex> g(mu);
    finished[id.v] = n;
  }
};

// generation.rs:195-253 subsample_children_into
static void subsample_children_into(const BuildCtx& ctx, NodeId node_id, FinishedSink* sink) {
  RawNodeWriter parent_writer(ctx, node_id);
  for (uint8_t i = 0; i < 8; ++i) {
    NodeId child_id = node_id.get_child_id(i);
    int64_t num_points;
    if (!number_of_points(ctx, child_id, &num_points)) continue;
    NodeIterator node_iterator(ctx, child_id, (size_t)num_points, NUM_POINTS_PER_BATCH);
    PointsBatch batch, b;
    // generation.rs:220 `.next().unwrap()` — a zero-byte .rgb cannot exist (files are removed).
    node_iterator.next(&batch);
    while (node_iterator.next(&b)) batch.append(b);
    size_t n = batch.len();
    std::vector<uint8_t> in_parent(n);
    for (size_t k = 0; k < n; ++k) in_parent[k] = (k % 8 == 0) ? 1 : 0;  // :224-229
    PointsBatch parent_batch, child_batch;
    batch.retain_into(in_parent, 1, &parent_batch);
    batch.retain_into(in_parent, 0, &child_batch);
    RawNodeWriter child_writer(ctx, child_id);
    parent_writer.write(parent_batch);
    child_writer.write(child_batch);
    sink->send(child_id, child_writer.num_written());
  }
  if (node_id.level() == 0) sink->send(node_id, parent_writer.num_written());  // :246-251
}

// ------------------------------------------------------------------------------------------------
// meta.pb — hand-rolled proto3 wire format (proto.proto:58-149; octree/mod.rs:87-99;
// node.rs:101-106,260-270). Field order follows the declaration order in the .proto, zero-valued
// scalars are omitted, sub-messages are always emitted.
// ------------------------------------------------------------------------------------------------
static void pb_varint(std::vector<uint8_t>& o, uint64_t v) {
  while (v >= 0x80) {
    o.push_back((uint8_t)(v | 0x80));
    v >>= 7;
  }
  o.push_back((uint8_t)v);
}
static void pb_tag(std::vector<uint8_t>& o, int field, int wt) { pb_varint(o, ((uint64_t)field << 3) | (uint64_t)wt); }
static void pb_double(std::vector<uint8_t>& o, int field, double d) {
  if (d == 0.) return;  // proto3 default omitted (also -0.0, as rust-protobuf's `!= 0.`)
  pb_tag(o, field, 1);
  uint64_t u;
  std::memcpy(&u, &d, 8);
  for (int i = 0; i < 8; ++i) o.push_back((uint8_t)(u >> (8 * i)));
}
static void pb_bytes(std::vector<uint8_t>& o, int field, const std::vector<uint8_t>& b) {
  pb_tag(o, field, 2);
  pb_varint(o, b.size());
  o.insert(o.end(), b.begin(), b.end());
}
static std::vector<uint8_t> pb_vec3d(const double v[3]) {
  std::vector<uint8_t> o;
  pb_double(o, 1, v[0]);
  pb_double(o, 2, v[1]);
  pb_double(o, 3, v[2]);
  return o;
}

struct MetaNode {
  NodeId id;
  int64_t num_points;
  Enc enc;
};
struct MetaData {
  int version = 0;
  Aabb bbox{};
  double resolution = 0;
  std::vector<MetaNode> nodes;
};

static std::vector<uint8_t> encode_meta(const MetaData& m) {
  std::vector<uint8_t> octree;
  pb_double(octree, 2, m.resolution);
  for (const MetaNode& n : m.nodes) {
    std::vector<uint8_t> node, id;
    if (n.enc != 0) {
      pb_tag(node, 2, 0);
      pb_varint(node, (uint64_t)n.enc);
    }
    if (n.num_points != 0) {
      pb_tag(node, 3, 0);
      pb_varint(node, (uint64_t)n.num_points);
    }
    if (n.id.high() != 0) {
      pb_tag(id, 3, 0);
      pb_varint(id, n.id.high());
    }
    if (n.id.low() != 0) {
      pb_tag(id, 4, 0);
      pb_varint(id, n.id.low());
    }
    pb_bytes(node, 4, id);
    pb_bytes(octree, 3, node);
  }
  std::vector<uint8_t> cuboid;
  pb_bytes(cuboid, 3, pb_vec3d(m.bbox.mn));
  pb_bytes(cuboid, 4, pb_vec3d(m.bbox.mx));
  std::vector<uint8_t> out;
  pb_tag(out, 1, 0);
  pb_varint(out, (uint64_t)m.version);
  pb_bytes(out, 4, cuboid);
  pb_bytes(out, 6, octree);
  return out;
}

struct PbReader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  uint64_t varint() {
    uint64_t v = 0;
    int s = 0;
    while (p < end) {
      uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << s;
      if (!(b & 0x80)) return v;
      s += 7;
      if (s > 63) break;
    }
    ok = false;
    return 0;
  }
  double f64() {
    if (end - p < 8) {
      ok = false;
      return 0;
    }
    uint64_t u = get_le(p, 8);
    p += 8;
    double d;
    std::memcpy(&d, &u, 8);
    return d;
  }
  PbReader sub() {
    uint64_t len = varint();
    if ((uint64_t)(end - p) < len) {
      ok = false;
      len = 0;
    }
    PbReader r{p, p + len};
    p += len;
    return r;
  }
  void skip(int wt) {
    switch (wt) {
      case 0: varint(); break;
      case 1: if (end - p >= 8) p += 8; else ok = false; break;
      case 2: sub(); break;
      case 5: if (end - p >= 4) p += 4; else ok = false; break;
      default: ok = false;
    }
  }
};

static void parse_vec3d(PbReader r, double v[3]) {
  v[0] = v[1] = v[2] = 0.;
  while (r.p < r.end && r.ok) {
    uint64_t t = r.varint();
    int f = (int)(t >> 3), wt = (int)(t & 7);
    if (wt == 1 && f >= 1 && f <= 3) v[f - 1] = r.f64();
    else r.skip(wt);
  }
}

static bool decode_meta(const std::vector<uint8_t>& buf, MetaData* m) {
  PbReader r{buf.data(), buf.data() + buf.size()};
  while (r.p < r.end && r.ok) {
    uint64_t t = r.varint();
    int f = (int)(t >> 3), wt = (int)(t & 7);
    if (f == 1 && wt == 0) m->version = (int)r.varint();
    else if (f == 4 && wt == 2) {
      PbReader c = r.sub();
      while (c.p < c.end && c.ok) {
        uint64_t ct = c.varint();
        int cf = (int)(ct >> 3), cw = (int)(ct & 7);
        if (cf == 3 && cw == 2) parse_vec3d(c.sub(), m->bbox.mn);
        else if (cf == 4 && cw == 2) parse_vec3d(c.sub(), m->bbox.mx);
        else c.skip(cw);
      }
    } else if (f == 6 && wt == 2) {
      PbReader o = r.sub();
      while (o.p < o.end && o.ok) {
        uint64_t ot = o.varint();
        int of = (int)(ot >> 3), ow = (int)(ot & 7);
        if (of == 2 && ow == 1) m->resolution = o.f64();
        else if (of == 3 && ow == 2) {
          PbReader n = o.sub();
          MetaNode mn{NodeId{0}, 0, ENC_INVALID};
          uint64_t hi = 0, lo = 0;
          while (n.p < n.end && n.ok) {
            uint64_t nt = n.varint();
            int nf = (int)(nt >> 3), nw = (int)(nt & 7);
            if (nf == 2 && nw == 0) mn.enc = (Enc)n.varint();
            else if (nf == 3 && nw == 0) mn.num_points = (int64_t)n.varint();
            else if (nf == 4 && nw == 2) {
              PbReader i = n.sub();
              while (i.p < i.end && i.ok) {
                uint64_t it = i.varint();
                int iff = (int)(it >> 3), iw = (int)(it & 7);
                if (iff == 3 && iw == 0) hi = i.varint();
                else if (iff == 4 && iw == 0) lo = i.varint();
                else i.skip(iw);
              }
            } else n.skip(nw);
          }
          mn.id = NodeId::from_high_low(hi, lo);
          m->nodes.push_back(mn);
        } else o.skip(ow);
      }
    } else r.skip(wt);
  }
  return r.ok;
}

// ------------------------------------------------------------------------------------------------
// Result object: the finished octree as (meta, per-node file bytes). Nodes are ordered by
// (level, index) — the reference's own order is nondeterministic (generation.rs:331,353-360; F6).
// ------------------------------------------------------------------------------------------------
struct ResultNode {
  NodeId id;
  int64_t num_points = 0;
  Enc enc = ENC_INVALID;
  std::vector<uint8_t> xyz, rgb, intensity;
  bool has_xyz = false, has_rgb = false, has_intensity = false;
};
struct Result {
  int version = 0;
  Aabb bbox{};
  double resolution = 0;
  std::vector<ResultNode> nodes;
  std::string error;
  // closed-form build only: max over all stored points and axes of |decode(stored code) - source coordinate|
  // (the "max abs position error" BASELINE config 5 asks for); -1 when not computed
  double max_abs_position_error = -1.0;
};

static void sort_nodes(Result* r) {
  std::sort(r->nodes.begin(), r->nodes.end(), [](const ResultNode& a, const ResultNode& b) {
    if (a.id.level() != b.id.level()) return a.id.level() < b.id.level();
    return a.id.index() < b.id.index();
  });
}

static Result* load_result(Backend& be) {
  Result* r = new Result();
  std::vector<uint8_t> buf;
  if (!be.read_all("meta.pb", &buf)) {
    r->error = "meta.pb not found";
    return r;
  }
  MetaData m;
  if (!decode_meta(buf, &m)) {
    r->error = "meta.pb does not parse";
    return r;
  }
  r->version = m.version;
  r->bbox = m.bbox;
  r->resolution = m.resolution;
  for (const MetaNode& mn : m.nodes) {
    ResultNode n;
    n.id = mn.id;
    n.num_points = mn.num_points;
    n.enc = mn.enc;
    std::string stem = mn.id.to_string();
    n.has_xyz = be.read_all(stem + ".xyz", &n.xyz);
    n.has_rgb = be.read_all(stem + ".rgb", &n.rgb);
    n.has_intensity = be.read_all(stem + ".intensity", &n.intensity);
    r->nodes.push_back(std::move(n));
  }
  sort_nodes(r);
  return r;
}

// generation.rs:289-403 build_octree (literal restatement).
static void build_literal(Backend& be, double resolution, const Aabb& bbox, InputStream& input,
                          int num_threads) {
  BuildCtx ctx;
  ctx.be = &be;
  ctx.resolution = resolution;
  ctx.bbox = bbox;
  ctx.root_cube = cube_bounding(bbox);
  ctx.has_intensity = input.intensity != nullptr;
  ctx.batch_size = input.batch;
  if (be.disk) ::mkdir(be.dir.c_str(), 0777);  // :308 ignore errors

  LeafSink leaf_sink;
  if (num_threads < 1) num_threads = 1;
#pragma omp parallel num_threads(num_threads)
  {
#pragma omp single
    { split_node(ctx, NodeId::root(), input, &leaf_sink); }
  }  // implicit barrier == end of rayon::scope

  std::vector<NodeId> nodes_to_subsample = leaf_sink.leaves;
  uint8_t deepest_level = 0;
  for (NodeId id : nodes_to_subsample) deepest_level = std::max(deepest_level, id.level());
  FinishedSink finished;
  for (int current_level = deepest_level; current_level >= 1; --current_level) {
    std::vector<NodeId> rest;
    std::set<u128> parent_set;
    for (NodeId n : nodes_to_subsample) {
      if (n.level() == current_level) {
        NodeId p;
        n.parent_id(&p);
        parent_set.insert(p.v);
      } else rest.push_back(n);
    }
    std::vector<NodeId> parent_ids;
    for (u128 v : parent_set) parent_ids.push_back(NodeId{v});
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
    for (size_t i = 0; i < parent_ids.size(); ++i) subsample_children_into(ctx, parent_ids[i], &finished);
    nodes_to_subsample = rest;
    nodes_to_subsample.insert(nodes_to_subsample.end(), parent_ids.begin(), parent_ids.end());
  }
  MetaData m;
  m.version = CURRENT_VERSION;
  m.bbox = bbox;
  m.resolution = resolution;
  for (auto& kv : finished.finished) {
    NodeId id{kv.first};
    Cube c = id.find_bounding_cube(ctx.root_cube);
    m.nodes.push_back(MetaNode{id, kv.second, position_encoding(c.edge, resolution)});
  }
  std::vector<uint8_t> buf = encode_meta(m);
  DataWriter w(&be, "meta.pb");
  w.write(buf.data(), buf.size());
  if (buf.empty()) {
    // cannot happen (version is always written) — keep DataWriter from deleting a legit file.
  }
}

// ------------------------------------------------------------------------------------------------
// CLOSED-FORM formulation (what the HIP pipeline computes).
// ------------------------------------------------------------------------------------------------
struct LevelTable {
  Cube root;
  double resolution;
  int max_level;             // deepest level a node can have: first k>=1 with edge[k] <= resolution
  std::vector<double> edge;  // edge[k], k = 0..max_level
  std::vector<Enc> enc;      // enc[k]
};

static LevelTable make_level_table(const Aabb& bbox, double resolution, int cap) {
  LevelTable t;
  t.root = cube_bounding(bbox);
  t.resolution = resolution;
  double e = t.root.edge;
  t.edge.push_back(e);
  t.enc.push_back(position_encoding(e, resolution));
  int k = 0;
  while (k < cap) {
    ++k;
    e /= 2.;  // node.rs:161
    t.edge.push_back(e);
    t.enc.push_back(position_encoding(e, resolution));
    if (e <= resolution) break;  // generation.rs:137 — such a node is never split again
  }
  t.max_level = k;
  return t;
}

// One step of the per-point chain: digit at the current cube, child cube, encode into
// the child cube with the child's encoding, decode back. p and mn are updated in place.
static inline uint8_t chain_step(const LevelTable& t, int k /*child level*/, double p[3], double mn[3],
                                 uint64_t code[3]) {
  Cube cur;
  cur.mn[0] = mn[0];
  cur.mn[1] = mn[1];
  cur.mn[2] = mn[2];
  cur.edge = t.edge[k - 1];
  uint8_t d = child_index_from_bounding_cube(cur, p);
  double e = t.edge[k];
  mn[0] += (double)((d >> 2) & 1) * e;  // node.rs:163-169
  mn[1] += (double)((d >> 1) & 1) * e;
  mn[2] += (double)(d & 1) * e;
  for (int a = 0; a < 3; ++a) {
    code[a] = encode_coord(t.enc[k], p[a], mn[a], e);
    p[a] = decode_coord(t.enc[k], code[a], mn[a], e);
  }
  return d;
}

// Routed input of the multi-rank build (tests only): a point given as its level-1 chain state — octant digit and the
// raw level-1 codes — instead of raw coordinates. chain_start() puts (p, mn, code) into the state chain_step() leaves
// behind after level 1 and returns the next level to run (2), or 1 for raw coordinates.
struct RoutedIn {
  const uint8_t* oct;
  const uint32_t* c[3];
};
static inline int chain_start(const LevelTable& t, const RoutedIn* r, size_t i, const double* x, const double* y,
                              const double* z, double p[3], double mn[3], uint64_t code[3], unsigned* d1) {
  mn[0] = t.root.mn[0];
  mn[1] = t.root.mn[1];
  mn[2] = t.root.mn[2];
  code[0] = code[1] = code[2] = 0;
  *d1 = 0;
  if (!r) {
    p[0] = x[i];
    p[1] = y[i];
    p[2] = z[i];
    return 1;
  }
  const unsigned d = r->oct[i];
  const double e = t.edge[1];
  mn[0] += (double)((d >> 2) & 1) * e;  // node.rs:163-169, as in chain_step
  mn[1] += (double)((d >> 1) & 1) * e;
  mn[2] += (double)(d & 1) * e;
  for (int a = 0; a < 3; ++a) {
    code[a] = r->c[a][i];
    p[a] = decode_coord(t.enc[1], code[a], mn[a], e);
  }
  *d1 = d;
  return 2;
}

// Path digits of one point for levels 1..nlevels, 3 bits per level, level 1 most significant.
static inline u128 chain_key(const LevelTable& t, int nlevels, const double p0[3]) {
  double p[3] = {p0[0], p0[1], p0[2]};
  double mn[3] = {t.root.mn[0], t.root.mn[1], t.root.mn[2]};
  uint64_t code[3];
  u128 key = 0;
  for (int k = 1; k <= nlevels; ++k) key = (key << 3) | chain_step(t, k, p, mn, code);
  return key;
}

struct CNode {
  NodeId id;
  int level;
  bool leaf;
  int child[8];
  int parent;
  std::vector<uint32_t> pre;  // indices of input points, in this node's stream order
  std::vector<uint8_t> origin_level;  // for inner nodes: leaf level each pre[] entry came from
};

// Multi-rank variant (tests of the sharded build only): `force_mask` splits the listed level-1 nodes whatever their
// local count, `streams_out` (8 level-1 + 64 level-2 stream lengths + split mask) reports the local top of the tree,
// and `layout` (root_points, l1_stream[8], l1_offset[8], l2_offset[64]) lays the root and the level-1 nodes out at their
// GLOBAL size: this rank's points land in their global slots, all other slots stay zero.
struct TopLayout {
  uint64_t root_points;
  uint64_t l1_stream[8];
  uint64_t l1_offset[8];
  uint64_t l2_offset[64];
};

static Result* build_closed(const Aabb& bbox, double resolution, size_t n, const double* x, const double* y,
                            const double* z, const uint8_t* rgb, const float* intensity, int num_threads,
                            unsigned force_mask = 0, const TopLayout* layout = nullptr, uint64_t* streams_out = nullptr,
                            const RoutedIn* routed = nullptr) {
  if (streams_out) std::memset(streams_out, 0, sizeof(uint64_t) * 73);
  Result* r = new Result();
  r->version = CURRENT_VERSION;
  r->bbox = bbox;
  r->resolution = resolution;
  if (n == 0) return r;  // generation.rs:325-330: no leaves -> no finished nodes
  if (num_threads < 1) num_threads = 1;
  const bool timing = std::getenv("PCVO_TIMING") != nullptr;
  double t_last = omp_get_wtime();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const double now = omp_get_wtime();
    std::fprintf(stderr, "[oracle closed] %-10s %.2f s\n", what, now - t_last);
    t_last = now;
  };
  LevelTable t = make_level_table(bbox, resolution, 40);
  int nlev = t.max_level;
  std::vector<u128> keys(n);
#pragma omp parallel for schedule(static) num_threads(num_threads)
  for (size_t i = 0; i < n; ++i) {
    double p[3], mn[3];
    uint64_t code[3];
    unsigned d1;
    const int k0 = chain_start(t, routed, i, x, y, z, p, mn, code, &d1);
    u128 key = d1;
    for (int k = k0; k <= nlev; ++k) key = (key << 3) | chain_step(t, k, p, mn, code);
    if (k0 > nlev) key = d1;
    keys[i] = key;
  }
  lap("keys");
  // Top-down stable split (generation.rs:58-193 without the files). The index lists of all nodes of a level live in ONE
  // array, every node a contiguous range of it; a level is split by counting the next digit per range, laying the eight
  // children out inside the parent's range and scattering into a second array (ping-pong per level) — no per-node
  // allocation. The order inside every child == the order inside the parent (the stable `retain` of
  // generation.rs:84-90). A leaf copies its range once; ranges are split in parallel over nodes, big ranges by chunks.
  std::vector<CNode> nodes;
  nodes.reserve(1024);
  std::vector<uint32_t> buf_a(n), buf_b(n);
#pragma omp parallel for schedule(static) num_threads(num_threads)
  for (size_t i = 0; i < n; ++i) buf_a[i] = (uint32_t)i;
  struct Range {
    size_t lo, hi;
  };
  std::vector<Range> range;  // per node: its list inside the buffer of its level
  {
    CNode root;
    root.id = NodeId::root();
    root.level = 0;
    root.leaf = false;  // the root is always split (generation.rs:312-323)
    root.parent = -1;
    for (int c = 0; c < 8; ++c) root.child[c] = -1;
    nodes.push_back(std::move(root));
    range.push_back(Range{0, n});
  }
  uint32_t* src = buf_a.data();
  uint32_t* dst = buf_b.data();
  for (size_t level_begin = 0; level_begin < nodes.size();) {
    const size_t level_end = nodes.size();
    const size_t width = level_end - level_begin;
    const int lvl = nodes[level_begin].level + 1;
    const int shift = 3 * (nlev - lvl);
    // work items: (node, chunk) — big ranges are cut into chunks so a handful of huge nodes still uses every thread
    struct Item {
      size_t node, lo, hi;
    };
    std::vector<Item> items;
    const size_t grain = std::max<size_t>(1u << 16, n / (size_t)(8 * num_threads) + 1);
    std::vector<size_t> first_item(width + 1, 0);
    for (size_t w = 0; w < width; ++w) {
      first_item[w] = items.size();
      if (nodes[level_begin + w].leaf) continue;
      const Range r = range[level_begin + w];
      for (size_t b0 = r.lo; b0 < r.hi; b0 += grain) items.push_back(Item{w, b0, std::min(r.hi, b0 + grain)});
    }
    first_item[width] = items.size();
    std::vector<size_t> cnt(items.size() * 8, 0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
    for (size_t it = 0; it < items.size(); ++it) {
      size_t local[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (size_t j = items[it].lo; j < items[it].hi; ++j) ++local[(unsigned)((keys[src[j]] >> shift) & 7)];
      for (int c = 0; c < 8; ++c) cnt[it * 8 + (size_t)c] = local[c];
    }
    // child c of a node starts at range.lo + (points of children < c); inside it the chunks follow each other
    std::vector<size_t> off(items.size() * 8, 0);
    std::vector<size_t> child_size(width * 8, 0);
    for (size_t w = 0; w < width; ++w) {
      if (nodes[level_begin + w].leaf) continue;
      size_t at = range[level_begin + w].lo;
      for (int c = 0; c < 8; ++c) {
        const size_t start = at;
        for (size_t it = first_item[w]; it < first_item[w + 1]; ++it) {
          off[it * 8 + (size_t)c] = at;
          at += cnt[it * 8 + (size_t)c];
        }
        child_size[w * 8 + (size_t)c] = at - start;
      }
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
    for (size_t it = 0; it < items.size(); ++it) {
      size_t at[8];
      for (int c = 0; c < 8; ++c) at[c] = off[it * 8 + (size_t)c];
      for (size_t j = items[it].lo; j < items[it].hi; ++j) {
        const uint32_t i = src[j];
        dst[at[(unsigned)((keys[i] >> shift) & 7)]++] = i;
      }
    }
    for (size_t w = 0; w < width; ++w) {
      const size_t qi = level_begin + w;
      if (nodes[qi].leaf) continue;
      size_t at = range[qi].lo;
      for (int c = 0; c < 8; ++c) {
        const size_t m = child_size[w * 8 + (size_t)c];
        if (m == 0) continue;
        CNode ch;
        ch.id = nodes[qi].id.get_child_id((uint8_t)c);
        ch.level = lvl;
        ch.parent = (int)qi;
        for (int k = 0; k < 8; ++k) ch.child[k] = -1;
        ch.leaf = !((int64_t)m > MAX_POINTS_PER_NODE && t.edge[lvl] > resolution);
        if (lvl == 1 && ((force_mask >> c) & 1u)) ch.leaf = false;
        nodes[qi].child[c] = (int)nodes.size();
        nodes.push_back(std::move(ch));
        range.push_back(Range{at, at + m});
        at += m;
      }
    }
    // the children that are leaves keep their list (it lives in `dst`, which the next level overwrites elsewhere only)
#pragma omp parallel for schedule(dynamic, 8) num_threads(num_threads)
    for (size_t qi = level_end; qi < nodes.size(); ++qi)
      if (nodes[qi].leaf) nodes[qi].pre.assign(dst + range[qi].lo, dst + range[qi].hi);
    std::swap(src, dst);
    level_begin = level_end;
  }
  buf_a = std::vector<uint32_t>();
  buf_b = std::vector<uint32_t>();
  lap("split");
  // Bottom-up promotion (generation.rs:195-253, 335-387): nodes[] is in BFS order, so reverse order
  // visits children before parents.
  for (size_t qi = 0; qi < nodes.size(); ++qi)
    if (nodes[qi].leaf) nodes[qi].origin_level.assign(nodes[qi].pre.size(), (uint8_t)nodes[qi].level);
  // final placement per node: (point index, leaf level it started at)
  std::vector<std::vector<uint32_t>> post_idx(nodes.size());
  std::vector<std::vector<uint8_t>> post_origin(nodes.size());
  // gpos[qi][j]: position of pre[j] in the node's GLOBAL stream (only differs from j for level <= 1 with a layout)
  std::vector<std::vector<uint64_t>> gpos(nodes.size());
  std::vector<std::vector<uint64_t>> post_slot(nodes.size());
  std::vector<int64_t> global_size(nodes.size(), -1);
  auto digit_of = [&](const CNode& nd) { return (unsigned)nd.id.child_index(); };
  for (size_t qi = nodes.size(); qi-- > 0;) {
    CNode& nd = nodes[qi];
    if (!nd.leaf) {
      nd.pre.clear();
      nd.origin_level.clear();
      for (int c = 0; c < 8; ++c) {
        if (nd.child[c] < 0) continue;
        CNode& ch = nodes[(size_t)nd.child[c]];
        if (streams_out && nd.level == 0) streams_out[c] = ch.pre.size();
        if (streams_out && nd.level == 1) streams_out[8 + digit_of(nd) * 8 + (unsigned)c] = ch.pre.size();
        if (streams_out && nd.level == 0 && !ch.leaf) streams_out[72] |= 1ull << c;
        const bool global = layout && nd.level <= 1;
        uint64_t base = 0;
        if (global) base = nd.level == 0 ? layout->l1_offset[c] : layout->l2_offset[digit_of(nd) * 8 + (unsigned)c];
        const std::vector<uint64_t>& cg = gpos[(size_t)nd.child[c]];
        for (size_t j = 0; j < ch.pre.size(); ++j) {
          const uint64_t g = cg.empty() ? j : cg[j];
          if (g % 8 != 0) continue;
          nd.pre.push_back(ch.pre[j]);
          nd.origin_level.push_back(ch.origin_level[j]);
          if (global) gpos[qi].push_back(base + g / 8);
        }
      }
    }
  }
  lap("promote");
#pragma omp parallel for schedule(dynamic, 16) num_threads(num_threads)
  for (size_t qi = 0; qi < nodes.size(); ++qi) {
    CNode& nd = nodes[qi];
    const bool global = layout && nd.level <= 1;
    if (global) {
      const uint64_t stream = nd.level == 0 ? layout->root_points : layout->l1_stream[digit_of(nd)];
      global_size[qi] = (int64_t)(nd.level == 0 ? stream : stream - (stream + 7) / 8);
    }
    for (size_t j = 0; j < nd.pre.size(); ++j) {
      const uint64_t g = gpos[qi].empty() ? j : gpos[qi][j];
      if (nd.level != 0 && g % 8 == 0) continue;  // promoted away; the root keeps everything
      post_idx[qi].push_back(nd.pre[j]);
      post_origin[qi].push_back(nd.origin_level[j]);
      if (global) post_slot[qi].push_back(nd.level == 0 ? g : g - g / 8 - 1);
    }
  }
  // Bytes: replay the chain down to the leaf level, then decode/encode upward, plus the one rewrite
  // every non-root node undergoes (generation.rs:230-238).
  lap("post");
  r->nodes.resize(nodes.size());
  double max_err = 0.0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads) reduction(max : max_err)
  for (size_t qi = 0; qi < nodes.size(); ++qi) {
    const CNode& nd = nodes[qi];
    ResultNode& out = r->nodes[qi];
    out.id = nd.id;
    const size_t m_local = post_idx[qi].size();
    const size_t m = global_size[qi] >= 0 ? (size_t)global_size[qi] : m_local;
    out.num_points = (int64_t)m;
    Enc enc = t.enc[nd.level];
    out.enc = enc;
    int bpc = bytes_per_coordinate(enc);
    out.has_xyz = out.has_rgb = m > 0;
    out.has_intensity = m > 0 && intensity != nullptr;
    out.xyz.assign(m * 3 * (size_t)bpc, 0);
    out.rgb.assign(m * 3, 0);
    if (intensity) out.intensity.assign(m * 4, 0);
    for (size_t sl = 0; sl < m_local; ++sl) {
      const size_t s = global_size[qi] >= 0 ? (size_t)post_slot[qi][sl] : sl;
      uint32_t i = post_idx[qi][sl];
      int L = post_origin[qi][sl];
      double p[3], mn[3];
      double mins[48][3];
      uint64_t code[3];
      unsigned d1;
      mins[0][0] = t.root.mn[0];
      mins[0][1] = t.root.mn[1];
      mins[0][2] = t.root.mn[2];
      const int k0 = chain_start(t, routed, i, x, y, z, p, mn, code, &d1);
      if (k0 == 2) {
        mins[1][0] = mn[0];
        mins[1][1] = mn[1];
        mins[1][2] = mn[2];
      }
      for (int k = k0; k <= L; ++k) {
        chain_step(t, k, p, mn, code);
        mins[k][0] = mn[0];
        mins[k][1] = mn[1];
        mins[k][2] = mn[2];
      }
      // climb from leaf level L to the final level nd.level
      for (int k = L; k > nd.level; --k)
        for (int a = 0; a < 3; ++a) {
          double q = decode_coord(t.enc[k], code[a], mins[k][a], t.edge[k]);
          code[a] = encode_coord(t.enc[k - 1], q, mins[k - 1][a], t.edge[k - 1]);
        }
      if (nd.level != 0)  // child rewrite: enc(dec(b)) once
        for (int a = 0; a < 3; ++a) {
          double q = decode_coord(enc, code[a], mins[nd.level][a], t.edge[nd.level]);
          code[a] = encode_coord(enc, q, mins[nd.level][a], t.edge[nd.level]);
        }
      for (int a = 0; a < 3; ++a) put_le(&out.xyz[(3 * s + (size_t)a) * (size_t)bpc], code[a], bpc);
      if (!routed) {  // what a reader of this node file gets back (raw.rs:141-215) against the source coordinate
        const double src[3] = {x[i], y[i], z[i]};
        for (int a = 0; a < 3; ++a) {
          const double err = std::fabs(decode_coord(enc, code[a], mins[nd.level][a], t.edge[nd.level]) - src[a]);
          if (err > max_err) max_err = err;  // NaN sources compare false and are skipped
        }
      }
      out.rgb[3 * s] = rgb[3 * (size_t)i];
      out.rgb[3 * s + 1] = rgb[3 * (size_t)i + 1];
      out.rgb[3 * s + 2] = rgb[3 * (size_t)i + 2];
      if (intensity) std::memcpy(&out.intensity[4 * s], &intensity[i], 4);
    }
  }
  lap("bytes");
  r->max_abs_position_error = routed ? -1.0 : max_err;
  sort_nodes(r);
  return r;
}

}  // namespace pcvo

// ------------------------------------------------------------------------------------------------
// C ABI for ctypes (tests/, bench.py cpu_baseline leg, __graft_entry__.smoke()).
// ------------------------------------------------------------------------------------------------
using namespace pcvo;

extern "C" {

// ---- scalar helpers for the reference's known-answer tests ----
uint64_t pcvo_encode_coord(int enc, double value, double mn, double edge) {
  return encode_coord((Enc)enc, value, mn, edge);
}
double pcvo_decode_coord(int enc, uint64_t raw, double mn, double edge) { return decode_coord((Enc)enc, raw, mn, edge); }
int pcvo_position_encoding(double edge, double resolution) { return (int)position_encoding(edge, resolution); }
int pcvo_child_index(const double cube_min[3], double edge, const double p[3]) {
  Cube c{{cube_min[0], cube_min[1], cube_min[2]}, edge};
  return child_index_from_bounding_cube(c, p);
}
void pcvo_node_id_from_string(const char* s, uint64_t* hi, uint64_t* lo) {
  NodeId n = NodeId::from_string(s);
  *hi = n.high();
  *lo = n.low();
}
void pcvo_node_id_to_string(uint64_t hi, uint64_t lo, char* out, int cap) {
  std::string s = NodeId::from_high_low(hi, lo).to_string();
  snprintf(out, (size_t)cap, "%s", s.c_str());
}
int pcvo_node_id_parent(uint64_t hi, uint64_t lo, uint64_t* phi, uint64_t* plo) {
  NodeId p;
  if (!NodeId::from_high_low(hi, lo).parent_id(&p)) return 0;
  *phi = p.high();
  *plo = p.low();
  return 1;
}
int pcvo_node_id_child_index(uint64_t hi, uint64_t lo) { return NodeId::from_high_low(hi, lo).child_index(); }
void pcvo_node_id_child(uint64_t hi, uint64_t lo, int c, uint64_t* chi, uint64_t* clo) {
  NodeId n = NodeId::from_high_low(hi, lo).get_child_id((uint8_t)c);
  *chi = n.high();
  *clo = n.low();
}
void pcvo_find_bounding_cube(uint64_t hi, uint64_t lo, const double root_min[3], double root_edge, double out_min[3],
                             double* out_edge) {
  Cube root{{root_min[0], root_min[1], root_min[2]}, root_edge};
  Cube c = NodeId::from_high_low(hi, lo).find_bounding_cube(root);
  out_min[0] = c.mn[0];
  out_min[1] = c.mn[1];
  out_min[2] = c.mn[2];
  *out_edge = c.edge;
}
void pcvo_cube_bounding(const double bmin[3], const double bmax[3], double out_min[3], double* out_edge) {
  Aabb b{{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}};
  Cube c = cube_bounding(b);
  out_min[0] = c.mn[0];
  out_min[1] = c.mn[1];
  out_min[2] = c.mn[2];
  *out_edge = c.edge;
}

// ---- level table + chain keys (stage-level parity for the HIP kernels) ----
// Returns max_level; fills edge[0..max_level], enc[0..max_level] (cap entries available).
int pcvo_level_table(const double bmin[3], const double bmax[3], double resolution, int cap, double* edge, int* enc) {
  Aabb b{{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}};
  LevelTable t = make_level_table(b, resolution, cap);
  for (int k = 0; k <= t.max_level; ++k) {
    edge[k] = t.edge[k];
    enc[k] = (int)t.enc[k];
  }
  return t.max_level;
}
// 64-bit keys: digit of level k in bits [3*(21-k), 3*(21-k)+3); levels beyond nlevels are zero.
void pcvo_chain_keys64(const double bmin[3], const double bmax[3], double resolution, int nlevels, uint64_t n,
                       const double* x, const double* y, const double* z, uint64_t* keys, int num_threads) {
  Aabb b{{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}};
  LevelTable t = make_level_table(b, resolution, 40);
  if (nlevels > t.max_level) nlevels = t.max_level;
  if (nlevels > 21) nlevels = 21;
  if (num_threads < 1) num_threads = 1;
#pragma omp parallel for schedule(static) num_threads(num_threads)
  for (uint64_t i = 0; i < n; ++i) {
    double p[3] = {x[i], y[i], z[i]};
    u128 k = chain_key(t, nlevels, p);
    keys[i] = (uint64_t)(k << (3 * (21 - nlevels)));
  }
}

void pcvo_aabb(uint64_t n, const double* x, const double* y, const double* z, double bmin[3], double bmax[3]) {
  // generation.rs:256-270 find_bounding_box; aabb.rs:41-44 grow (inf/sup per component).
  if (n == 0) {
    for (int a = 0; a < 3; ++a) bmin[a] = bmax[a] = 0.;  // Aabb::zero()
    return;
  }
  const double* c[3] = {x, y, z};
  for (int a = 0; a < 3; ++a) {
    double lo = c[a][0], hi = c[a][0];
    for (uint64_t i = 1; i < n; ++i) {
      double v = c[a][i];
      if (v < lo) lo = v;
      if (v > hi) hi = v;
    }
    bmin[a] = lo;
    bmax[a] = hi;
  }
}

// ---- builders ----
// Literal build into a real directory (CPU baseline + directory parity). Returns 0.
int pcvo_build_literal_dir(const char* dir, double resolution, const double bmin[3], const double bmax[3], uint64_t n,
                           const double* x, const double* y, const double* z, const uint8_t* rgb,
                           const float* intensity, uint64_t batch_size, int num_threads) {
  Backend be;
  be.disk = true;
  be.dir = dir;
  Aabb b{{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}};
  InputStream in;
  in.x = x;
  in.y = y;
  in.z = z;
  in.rgb = rgb;
  in.intensity = intensity;
  in.n = n;
  in.batch = batch_size ? batch_size : NUM_POINTS_PER_BATCH;
  build_literal(be, resolution, b, in, num_threads);
  return 0;
}

// Literal build through the in-memory file map; returns a Result handle.
void* pcvo_build_literal_mem(double resolution, const double bmin[3], const double bmax[3], uint64_t n,
                             const double* x, const double* y, const double* z, const uint8_t* rgb,
                             const float* intensity, uint64_t batch_size, int num_threads) {
  Backend be;
  Aabb b{{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}};
  InputStream in;
  in.x = x;
  in.y = y;
  in.z = z;
  in.rgb = rgb;
  in.intensity = intensity;
  in.n = n;
  in.batch = batch_size ? batch_size : NUM_POINTS_PER_BATCH;
  build_literal(be, resolution, b, in, num_threads);
  return load_result(be);
}

void* pcvo_build_closed(double resolution, const double bmin[3], const double bmax[3], uint64_t n, const double* x,
                        const double* y, const double* z, const uint8_t* rgb, const float* intensity,
                        int num_threads) {
  Aabb b{{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}};
  return build_closed(b, resolution, n, x, y, z, rgb, intensity, num_threads);
}

// One rank's share of a multi-rank build (see TopLayout). layout: 1 + 8 + 8 + 64 u64 or NULL; streams_out: 73 u64 or NULL.
void* pcvo_build_closed_shard(double resolution, const double bmin[3], const double bmax[3], uint64_t n, const double* x,
                              const double* y, const double* z, const uint8_t* rgb, const float* intensity,
                              int num_threads, unsigned force_mask, const uint64_t* layout, uint64_t* streams_out,
                              const uint8_t* r_oct, const uint32_t* r_cx, const uint32_t* r_cy, const uint32_t* r_cz) {
  RoutedIn ri{r_oct, {r_cx, r_cy, r_cz}};
  Aabb b{{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}};
  TopLayout tl;
  if (layout) {
    tl.root_points = layout[0];
    for (int c = 0; c < 8; ++c) {
      tl.l1_stream[c] = layout[1 + c];
      tl.l1_offset[c] = layout[9 + c];
    }
    for (int i = 0; i < 64; ++i) tl.l2_offset[i] = layout[17 + i];
  }
  return build_closed(b, resolution, n, x, y, z, rgb, intensity, num_threads, force_mask, layout ? &tl : nullptr, streams_out,
                      r_oct ? &ri : nullptr);
}

// Level-1 chain state of raw points: octant digit + raw level-1 codes (what crosses the exchange of the multi-rank build).
void pcvo_chain_state1(const double bmin[3], const double bmax[3], double resolution, uint64_t n, const double* x,
                       const double* y, const double* z, uint8_t* oct, uint32_t* cx, uint32_t* cy, uint32_t* cz) {
  Aabb b{{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}};
  LevelTable t = make_level_table(b, resolution, 40);
  for (uint64_t i = 0; i < n; ++i) {
    double p[3] = {x[i], y[i], z[i]};
    double mn[3] = {t.root.mn[0], t.root.mn[1], t.root.mn[2]};
    uint64_t code[3];
    oct[i] = chain_step(t, 1, p, mn, code);
    cx[i] = (uint32_t)code[0];
    cy[i] = (uint32_t)code[1];
    cz[i] = (uint32_t)code[2];
  }
}

// Load an octree directory (ours or the oracle's) for comparison.
void* pcvo_load_dir(const char* dir) {
  Backend be;
  be.disk = true;
  be.dir = dir;
  return load_result(be);
}

double pcvo_result_max_abs_position_error(void* h) { return ((Result*)h)->max_abs_position_error; }
const char* pcvo_result_error(void* h) { return ((Result*)h)->error.c_str(); }
int pcvo_result_version(void* h) { return ((Result*)h)->version; }
double pcvo_result_resolution(void* h) { return ((Result*)h)->resolution; }
void pcvo_result_bbox(void* h, double bmin[3], double bmax[3]) {
  Result* r = (Result*)h;
  for (int a = 0; a < 3; ++a) {
    bmin[a] = r->bbox.mn[a];
    bmax[a] = r->bbox.mx[a];
  }
}
uint64_t pcvo_result_num_nodes(void* h) { return ((Result*)h)->nodes.size(); }
void pcvo_result_node(void* h, uint64_t i, uint64_t* hi, uint64_t* lo, int64_t* num_points, int* enc, int* level,
                      int* has_files) {
  const ResultNode& n = ((Result*)h)->nodes[i];
  *hi = n.id.high();
  *lo = n.id.low();
  *num_points = n.num_points;
  *enc = (int)n.enc;
  *level = n.id.level();
  *has_files = (n.has_xyz ? 1 : 0) | (n.has_rgb ? 2 : 0) | (n.has_intensity ? 4 : 0);
}
// which: 0 xyz, 1 rgb, 2 intensity
const uint8_t* pcvo_result_node_data(void* h, uint64_t i, int which, uint64_t* len) {
  const ResultNode& n = ((Result*)h)->nodes[i];
  const std::vector<uint8_t>& v = which == 0 ? n.xyz : (which == 1 ? n.rgb : n.intensity);
  *len = v.size();
  return v.data();
}
void pcvo_result_free(void* h) { delete (Result*)h; }

// meta.pb encode/decode round trip helpers (used to validate pcv's own writer).
uint64_t pcvo_meta_encode(int version, const double bmin[3], const double bmax[3], double resolution, uint64_t nn,
                          const uint64_t* hi, const uint64_t* lo, const int64_t* num_points, const int* enc,
                          uint8_t* out, uint64_t cap) {
  MetaData m;
  m.version = version;
  for (int a = 0; a < 3; ++a) {
    m.bbox.mn[a] = bmin[a];
    m.bbox.mx[a] = bmax[a];
  }
  m.resolution = resolution;
  for (uint64_t i = 0; i < nn; ++i) m.nodes.push_back(MetaNode{NodeId::from_high_low(hi[i], lo[i]), num_points[i], (Enc)enc[i]});
  std::vector<uint8_t> b = encode_meta(m);
  if (b.size() <= cap) std::memcpy(out, b.data(), b.size());
  return b.size();
}

int pcvo_num_procs(void) { return omp_get_num_procs(); }
void pcvo_set_max_points_per_node(int64_t v) { MAX_POINTS_PER_NODE = v; }
int64_t pcvo_get_max_points_per_node(void) { return MAX_POINTS_PER_NODE; }

}  // extern "C"
