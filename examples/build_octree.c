/* build_octree.c — the reference's `build_octree` binary (src/bin/build_octree.rs:41-53) on top of the C ABI, in plain
 * C11: PLY -> octree directory. Proves that include/pcv_hip.h is a C header and shows the call sequence a non-Python
 * host uses:  pcv_ply_read -> pcv_build_octree(PCV_BUILD_COMPUTE_BBOX) -> pcv_octree_write_dir
 * == build_octree_from_file(output_directory, resolution, input, &["color", "intensity"]) (generation.rs:272-287).
 *
 *   build_octree <input.ply> --output-directory <dir> [--resolution 0.001] [--num-threads N (ignored: GPU build)]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcv_layout_check.h"

static int usage(void) {
  fprintf(stderr, "usage: build_octree <input.ply> --output-directory <dir> [--resolution 0.001] [--num-threads N]\n");
  return 2;
}

int main(int argc, char** argv) {
  const char* input = NULL;
  const char* outdir = NULL;
  double resolution = 0.001; /* src/bin/build_octree.rs:33 */
  for (int i = 1; i < argc; ++i) {
    if (strcmp(argv[i], "--output-directory") == 0 && i + 1 < argc) outdir = argv[++i];
    else if (strcmp(argv[i], "--resolution") == 0 && i + 1 < argc) resolution = atof(argv[++i]);
    else if (strcmp(argv[i], "--num-threads") == 0 && i + 1 < argc) ++i; /* rayon pool size: nothing to size here */
    else if (argv[i][0] != '-' && !input) input = argv[i];
    else return usage();
  }
  if (!input || !outdir) return usage();

  char err[512] = {0};
  pcv_ply* ply = NULL;
  int rc = pcv_ply_read(input, &ply, err, sizeof(err));
  if (rc != PCV_OK) {
    fprintf(stderr, "cannot read %s: %s\n", input, err);
    return 1;
  }
  pcv_points pts;
  pcv_ply_points(ply, &pts);
  if (!pts.color) { /* on_disk.rs:20-22: the octree format always carries colour */
    fprintf(stderr, "%s has no red/green/blue properties\n", input);
    pcv_ply_free(ply);
    return 1;
  }
  if (!pts.intensity) { /* the reference binary asks for "intensity" and panics without it */
    fprintf(stderr, "%s has no 'intensity' property (the reference build_octree requires it)\n", input);
    pcv_ply_free(ply);
    return 1;
  }
  pcv_ctx* ctx = NULL;
  if ((rc = pcv_ctx_create(0, NULL, &ctx)) != PCV_OK) {
    fprintf(stderr, "no HIP device (pcv_ctx_create: %d); there is no CPU fallback\n", rc);
    pcv_ply_free(ply);
    return 1;
  }
  pcv_build_params params;
  memset(&params, 0, sizeof(params));
  params.resolution = resolution;
  params.flags = PCV_BUILD_COMPUTE_BBOX; /* find_bounding_box on the device (generation.rs:256-270) */
  pcv_octree* tree = NULL;
  rc = pcv_build_octree(ctx, &params, &pts, &tree);
  if (rc == PCV_OK) rc = pcv_octree_write_dir(tree, outdir);
  if (rc != PCV_OK) fprintf(stderr, "build failed (%d): %s\n", rc, pcv_last_error(ctx));
  else
    printf("%llu points -> %llu nodes in %s\n", (unsigned long long)pcv_octree_num_points(tree),
           (unsigned long long)pcv_octree_num_nodes(tree), outdir);
  pcv_octree_free(tree);
  pcv_ply_free(ply);
  pcv_ctx_destroy(ctx);
  return rc == PCV_OK ? 0 : 1;
}
