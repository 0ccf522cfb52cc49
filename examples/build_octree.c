/* build_octree.c — the reference's `build_octree` binary (src/bin/build_octree.rs:41-53) on top of the C ABI, in plain
 * C11: PLY -> octree directory. Proves that include/pcv_hip.h is a C header and shows the call sequence a non-Python
 * host uses:  pcv_build_octree_from_ply (decode + bounding box on the device) -> pcv_octree_write_dir
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

  pcv_ctx* ctx = NULL;
  int rc;
  if ((rc = pcv_ctx_create(0, NULL, &ctx)) != PCV_OK) {
    fprintf(stderr, "no HIP device (pcv_ctx_create: %d); there is no CPU fallback\n", rc);
    return 1;
  }
  pcv_build_params params;
  memset(&params, 0, sizeof(params));
  params.resolution = resolution;
  /* build_octree_from_file (generation.rs:272-287): the vertex records go to the device as they are in the file and
   * are decoded there; find_bounding_box runs on the device too. Like the reference binary (src/bin/build_octree.rs:47-52)
   * this asks for "intensity" and fails on a file without it. */
  pcv_octree* tree = NULL;
  rc = pcv_build_octree_from_ply(ctx, &params, input, 1, &tree);
  if (rc == PCV_OK) rc = pcv_octree_write_dir(tree, outdir);
  if (rc != PCV_OK) fprintf(stderr, "build of %s failed (%d): %s\n", input, rc, pcv_last_error(ctx));
  else
    printf("%llu points -> %llu nodes in %s\n", (unsigned long long)pcv_octree_num_points(tree),
           (unsigned long long)pcv_octree_num_nodes(tree), outdir);
  pcv_octree_free(tree);
  pcv_ctx_destroy(ctx);
  return rc == PCV_OK ? 0 : 1;
}
