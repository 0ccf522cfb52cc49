// pcv_ply.cpp — binary little-endian PLY ingest into SoA host arrays (host side, no GPU work).
//
// Restates PlyIterator (reference src/read_write/ply.rs:126-221 header, :328-455 property readers, :457-511 batch
// assembly): element `vertex` with scalar properties; x / y / z of any scalar type are cast to f64 and the header's
// `comment offset: x y z` is added; red|r, green|g, blue|b (uchar) form the colour; `intensity` (float) is kept;
// a|alpha and every other property are skipped; list properties are ignored. One pass over the file fills the SoA
// arrays the build consumes directly (the reference reads the file twice: bounding box, then batches).
// Not supported (rejected with PCV_E_INVALID): ascii / big-endian bodies (the reference panics on them too) and
// 64-bit integer properties (the reference advances its cursor by 4 bytes for them, ply.rs:278-283).
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/pcv_hip.h"
#include "pcv_ply_layout.h"

struct pcv_ply {
  std::vector<double> x, y, z;
  std::vector<uint8_t> rgb;
  std::vector<float> intensity;
  bool has_color = false, has_intensity = false;
  double offset[3] = {0, 0, 0};
};

namespace {

enum Type { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_BAD };  // == PcvPlyType (pcv_ply_layout.h)
static_assert(T_I8 == (int)PCV_PLY_I8 && T_U8 == (int)PCV_PLY_U8 && T_I16 == (int)PCV_PLY_I16 && T_U16 == (int)PCV_PLY_U16 &&
                  T_I32 == (int)PCV_PLY_I32 && T_U32 == (int)PCV_PLY_U32 && T_F32 == (int)PCV_PLY_F32 && T_F64 == (int)PCV_PLY_F64,
              "the host parser and the device decode share the type numbering");
Type parse_type(const std::string& s) {  // ply.rs:62-80 DataType::from_str
  if (s == "char" || s == "int8") return T_I8;
  if (s == "uchar" || s == "uint8") return T_U8;
  if (s == "short" || s == "int16") return T_I16;
  if (s == "ushort" || s == "uint16") return T_U16;
  if (s == "int" || s == "int32") return T_I32;
  if (s == "uint" || s == "uint32") return T_U32;
  if (s == "float" || s == "float32") return T_F32;
  if (s == "double" || s == "float64") return T_F64;
  return T_BAD;
}
int type_size(Type t) {
  switch (t) {
    case T_I8: case T_U8: return 1;
    case T_I16: case T_U16: return 2;
    case T_I32: case T_U32: case T_F32: return 4;
    default: return 8;
  }
}
double read_as_f64(Type t, const uint8_t* p) {
  switch (t) {
    case T_I8: return (double)(int8_t)p[0];
    case T_U8: return (double)p[0];
    case T_I16: { int16_t v; std::memcpy(&v, p, 2); return (double)v; }
    case T_U16: { uint16_t v; std::memcpy(&v, p, 2); return (double)v; }
    case T_I32: { int32_t v; std::memcpy(&v, p, 4); return (double)v; }
    case T_U32: { uint32_t v; std::memcpy(&v, p, 4); return (double)v; }
    case T_F32: { float v; std::memcpy(&v, p, 4); return (double)v; }
    default: { double v; std::memcpy(&v, p, 8); return v; }
  }
}

struct Prop {
  std::string name;
  Type type;
  int offset;
};

int fail(char* err, size_t cap, int code, const std::string& msg) {
  if (err && cap) snprintf(err, cap, "%s", msg.c_str());
  return code;
}

bool read_line(FILE* f, std::string* line) {
  line->clear();
  int c;
  while ((c = fgetc(f)) != EOF) {
    if (c == '\n') return true;
    line->push_back((char)c);
  }
  return !line->empty();
}

}  // namespace

// Header of a binary little-endian PLY (ply.rs:126-221) -> where the vertex records start and how one is laid out.
// Leaves `f` at the first vertex record. The header is untrusted: a vertex count the rest of the file cannot hold
// (negative, absurd) is rejected here, before anything is sized by it.
int pcv_ply_parse_header(FILE* f, PcvPlyLayout* lay, char* err, uint64_t errcap) {
  std::string line;
  if (!read_line(f, &line) || line.find("ply") != 0 || line.find_first_not_of(" \r\t", 3) != std::string::npos)
    return fail(err, errcap, PCV_E_INVALID, "Not a PLY file");
  bool have_format = false, little = false, in_vertex = false, have_vertex = false, ended = false;
  long long vertex_count = 0;
  bool vertex_first = true, seen_element = false;
  std::vector<Prop> props;
  int stride = 0;
  *lay = PcvPlyLayout();
  while (read_line(f, &line)) {
    std::istringstream ss(line);
    std::vector<std::string> e;
    std::string tok;
    while (ss >> tok) e.push_back(tok);
    if (e.empty()) return fail(err, errcap, PCV_E_INVALID, "Invalid line: " + line);
    if (e[0] == "format" && e.size() == 3) {
      if (e[2] != "1.0") return fail(err, errcap, PCV_E_INVALID, "Invalid version: " + e[2]);
      have_format = true;
      little = e[1] == "binary_little_endian";
      if (e[1] != "ascii" && e[1] != "binary_little_endian" && e[1] != "binary_big_endian")
        return fail(err, errcap, PCV_E_INVALID, "Invalid format: " + e[1]);
    } else if (e[0] == "element" && e.size() == 3) {
      in_vertex = e[1] == "vertex";
      if (in_vertex) {
        have_vertex = true;
        vertex_first = !seen_element;
        vertex_count = atoll(e[2].c_str());
      }
      seen_element = true;
    } else if (e[0] == "property") {
      if (!seen_element) return fail(err, errcap, PCV_E_INVALID, "property outside of element: " + line);
      if (e.size() == 5 && e[1] == "list") continue;  // list properties are not supported (ignored)
      if (e.size() != 3) return fail(err, errcap, PCV_E_INVALID, "Invalid line: " + line);
      if (in_vertex) {
        Type t = parse_type(e[1]);
        if (t == T_BAD) return fail(err, errcap, PCV_E_INVALID, "Invalid or unsupported data type: " + e[1]);
        props.push_back(Prop{e[2], t, stride});
        stride += type_size(t);
      }
    } else if (e[0] == "end_header") {
      ended = true;
      break;
    } else if (e[0] == "comment") {
      if (e.size() == 5 && e[1] == "offset:") {
        for (int a = 0; a < 3; ++a) lay->offset[a] = atof(e[2 + a].c_str());
      }
    } else {
      return fail(err, errcap, PCV_E_INVALID, "Invalid line: " + line);
    }
  }
  if (!ended || !have_format) return fail(err, errcap, PCV_E_INVALID, "No format specified");
  if (!have_vertex) return fail(err, errcap, PCV_E_INVALID, "Header does not have element 'vertex'");
  if (!little) return fail(err, errcap, PCV_E_INVALID, "Unsupported PLY format (only binary_little_endian bodies)");
  if (!vertex_first) return fail(err, errcap, PCV_E_INVALID, "element 'vertex' must be the first element");
  int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1, ii = -1;
  for (size_t k = 0; k < props.size(); ++k) {
    const std::string& nm = props[k].name;
    if (nm == "x") ix = (int)k;
    else if (nm == "y") iy = (int)k;
    else if (nm == "z") iz = (int)k;
    else if (nm == "r" || nm == "red") ir = (int)k;
    else if (nm == "g" || nm == "green") ig = (int)k;
    else if (nm == "b" || nm == "blue") ib = (int)k;
    else if (nm == "intensity" && props[k].type == T_F32) ii = (int)k;
  }
  if (ix < 0 || iy < 0 || iz < 0) return fail(err, errcap, PCV_E_INVALID, "PLY must contain properties 'x', 'y', 'z' for 'vertex'.");
  const bool has_color = ir >= 0 && ig >= 0 && ib >= 0;
  if (has_color && (props[ir].type != T_U8 || props[ig].type != T_U8 || props[ib].type != T_U8))
    return fail(err, errcap, PCV_E_INVALID, "colour properties must be uchar");
  if (vertex_count < 0) return fail(err, errcap, PCV_E_INVALID, "negative vertex count");
  const long body = ftell(f);
  long total = -1;
  if (body >= 0 && fseek(f, 0, SEEK_END) == 0) total = ftell(f);
  if (body < 0 || total < body || fseek(f, body, SEEK_SET) != 0) return fail(err, errcap, PCV_E_IO, "cannot measure the vertex data");
  if (stride <= 0 || (unsigned long long)vertex_count > (unsigned long long)(total - body) / (unsigned long long)stride)
    return fail(err, errcap, PCV_E_IO, "unexpected end of file in the vertex data");
  lay->vertex_count = vertex_count;
  lay->stride = stride;
  lay->body_offset = body;
  lay->x_type = props[ix].type, lay->x_off = props[ix].offset;
  lay->y_type = props[iy].type, lay->y_off = props[iy].offset;
  lay->z_type = props[iz].type, lay->z_off = props[iz].offset;
  lay->r_off = has_color ? props[ir].offset : -1;
  lay->g_off = has_color ? props[ig].offset : -1;
  lay->b_off = has_color ? props[ib].offset : -1;
  lay->i_off = ii >= 0 ? props[ii].offset : -1;
  return PCV_OK;
}

extern "C" int pcv_ply_read(const char* path, pcv_ply** out, char* err, uint64_t errcap) {
  if (!path || !out) return PCV_E_INVALID;
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return fail(err, errcap, PCV_E_IO, "Could not open input file.");
  PcvPlyLayout lay;
  int rc = pcv_ply_parse_header(f, &lay, err, errcap);
  if (rc != PCV_OK) {
    fclose(f);
    return rc;
  }
  pcv_ply* ply = new pcv_ply();
  auto bail = [&](int code, const std::string& msg) {
    delete ply;
    fclose(f);
    return fail(err, errcap, code, msg);
  };
  for (int a = 0; a < 3; ++a) ply->offset[a] = lay.offset[a];
  ply->has_color = lay.r_off >= 0;
  ply->has_intensity = lay.i_off >= 0;
  const int stride = lay.stride;
  // no allocation failure may unwind across the C ABI
  const size_t n = (size_t)lay.vertex_count;
  const size_t chunk_pts = 1 << 16;
  std::vector<uint8_t> buf;
  try {
    ply->x.resize(n);
    ply->y.resize(n);
    ply->z.resize(n);
    if (ply->has_color) ply->rgb.resize(3 * n);
    if (ply->has_intensity) ply->intensity.resize(n);
    buf.resize(chunk_pts * (size_t)stride);
  } catch (...) {
    return bail(PCV_E_OOM, "out of host memory for " + std::to_string(n) + " points");
  }
  for (size_t done = 0; done < n;) {
    const size_t m = std::min(chunk_pts, n - done);
    if (fread(buf.data(), (size_t)stride, m, f) != m) return bail(PCV_E_IO, "unexpected end of file in the vertex data");
    for (size_t k = 0; k < m; ++k) {
      const uint8_t* p = buf.data() + k * (size_t)stride;
      // ply.rs:488-493: cast to f64, then add the header offset
      ply->x[done + k] = read_as_f64((Type)lay.x_type, p + lay.x_off) + ply->offset[0];
      ply->y[done + k] = read_as_f64((Type)lay.y_type, p + lay.y_off) + ply->offset[1];
      ply->z[done + k] = read_as_f64((Type)lay.z_type, p + lay.z_off) + ply->offset[2];
      if (ply->has_color) {
        ply->rgb[3 * (done + k)] = p[lay.r_off];
        ply->rgb[3 * (done + k) + 1] = p[lay.g_off];
        ply->rgb[3 * (done + k) + 2] = p[lay.b_off];
      }
      if (ply->has_intensity) std::memcpy(&ply->intensity[done + k], p + lay.i_off, 4);
    }
    done += m;
  }
  fclose(f);
  *out = ply;
  return PCV_OK;
}

extern "C" uint64_t pcv_ply_num_points(const pcv_ply* p) { return p ? p->x.size() : 0; }

extern "C" int pcv_ply_points(const pcv_ply* p, pcv_points* out) {
  if (!p || !out) return PCV_E_INVALID;
  out->n = p->x.size();
  out->x = p->x.data();
  out->y = p->y.data();
  out->z = p->z.data();
  out->color = p->has_color ? p->rgb.data() : nullptr;
  out->color_stride = 3;
  out->intensity = p->has_intensity ? p->intensity.data() : nullptr;
  out->mem = PCV_MEM_HOST;
  return PCV_OK;
}

extern "C" void pcv_ply_free(pcv_ply* p) { delete p; }
