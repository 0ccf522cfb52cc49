// query_octree.cc — C++ consumer of include/pcv_hip.h for the query side: open an octree directory, ask for the nodes
// visible from a camera matrix (Octree::get_visible_nodes, src/octree/mod.rs:228-283) and stream the points of an
// axis-aligned box (PointCloud::nodes_in_location + FilteredIterator, src/iterator.rs:96-119,185-223) — the calls
// `sdl_viewer` / `point_cloud_client` make, from a compiled non-Python host.
//
//   query_octree <octree dir> [min_x min_y min_z max_x max_y max_z]
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "pcv_layout_check.h"

namespace {
struct CtxDeleter {
  void operator()(pcv_ctx* c) const { pcv_ctx_destroy(c); }
};
struct TreeDeleter {
  void operator()(pcv_octree* t) const { pcv_octree_free(t); }
};
struct ShapesDeleter {
  void operator()(pcv_shapes* s) const { pcv_shapes_free(s); }
};
}  // namespace

int main(int argc, char** argv) {
  if (argc != 2 && argc != 8) {
    std::fprintf(stderr, "usage: query_octree <octree dir> [min_x min_y min_z max_x max_y max_z]\n");
    return 2;
  }
  pcv_ctx* raw_ctx = nullptr;
  if (pcv_ctx_create(0, nullptr, &raw_ctx) != PCV_OK) {
    std::fprintf(stderr, "no HIP device; there is no CPU fallback\n");
    return 1;
  }
  std::unique_ptr<pcv_ctx, CtxDeleter> ctx(raw_ctx);
  pcv_octree* raw_tree = nullptr;
  if (pcv_octree_open_dir(ctx.get(), argv[1], &raw_tree) != PCV_OK) {
    std::fprintf(stderr, "%s\n", pcv_last_error(ctx.get()));
    return 1;
  }
  std::unique_ptr<pcv_octree, TreeDeleter> tree(raw_tree);
  double resolution = 0, bmin[3], bmax[3];
  int version = 0;
  pcv_octree_meta(tree.get(), &resolution, bmin, bmax, &version);
  std::printf("%llu nodes, %llu points, resolution %g\n", (unsigned long long)pcv_octree_num_nodes(tree.get()),
              (unsigned long long)pcv_octree_num_points(tree.get()), resolution);

  pcv_shape box{};
  box.kind = PCV_SHAPE_AABB;
  for (int a = 0; a < 3; ++a) {
    box.params[a] = argc == 8 ? std::atof(argv[2 + a]) : bmin[a];
    box.params[3 + a] = argc == 8 ? std::atof(argv[5 + a]) : bmin[a] + 0.5 * (bmax[a] - bmin[a]);
  }
  pcv_shapes* raw_shapes = nullptr;
  if (pcv_shapes_create(ctx.get(), &box, 1, &raw_shapes) != PCV_OK) {
    std::fprintf(stderr, "%s\n", pcv_last_error(ctx.get()));
    return 1;
  }
  std::unique_ptr<pcv_shapes, ShapesDeleter> shapes(raw_shapes);
  const uint64_t cap = pcv_octree_num_points(tree.get());
  std::vector<double> x(cap), y(cap), z(cap);
  std::vector<uint8_t> rgb(3 * cap);
  uint64_t count = 0;
  if (pcv_query_points(ctx.get(), shapes.get(), 0, tree.get(), nullptr, cap, PCV_MEM_HOST, x.data(), y.data(), z.data(),
                       rgb.data(), nullptr, &count) != PCV_OK) {
    std::fprintf(stderr, "%s\n", pcv_last_error(ctx.get()));
    return 1;
  }
  std::printf("%llu points inside the box\n", (unsigned long long)count);
  return 0;
}
