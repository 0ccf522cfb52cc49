/* ingest_batches.c — the reference's library entry `build_octree(dir, resolution, bounding_box, input, attributes)`
 * (src/octree/generation.rs:289-295) over the C ABI, in plain C11, with `input` arriving the way the reference's iterators
 * deliver it: one PointsBatch at a time (src/lib.rs:102-107: positions AoS, Vec<Point3<f64>>; colour Vec<Vector3<u8>>;
 * optional intensity Vec<f32>), 500 000 points per batch (src/lib.rs:52). The batches are read from raw files here — a real
 * host would hand over whatever its reader produced — and every batch is passed to pcv_ingest_append AS IT IS: no SoA
 * conversion, no whole-cloud buffer on the host.
 *
 *   ingest_batches <xyz.f64 (n x 3 doubles)> <rgb.u8 (n x 3 bytes)> <intensity.f32 | -> <n> <batch> <dir> [resolution]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcv_layout_check.h"

int main(int argc, char** argv) {
  if (argc < 7) {
    fprintf(stderr, "usage: ingest_batches <xyz.f64> <rgb.u8> <intensity.f32 | -> <n> <batch> <dir> [resolution]\n");
    return 2;
  }
  const unsigned long long n = strtoull(argv[4], NULL, 10), batch = strtoull(argv[5], NULL, 10);
  const int with_intensity = strcmp(argv[3], "-") != 0;
  const double resolution = argc > 7 ? atof(argv[7]) : 0.001;
  FILE* fx = fopen(argv[1], "rb");
  FILE* fc = fopen(argv[2], "rb");
  FILE* fi = with_intensity ? fopen(argv[3], "rb") : NULL;
  if (!fx || !fc || (with_intensity && !fi) || batch == 0) {
    fprintf(stderr, "cannot open the inputs\n");
    return 1;
  }
  /* one batch of host memory, whatever the size of the cloud */
  double* xyz = (double*)malloc((size_t)batch * 3 * sizeof(double));
  unsigned char* rgb = (unsigned char*)malloc((size_t)batch * 3);
  float* inten = with_intensity ? (float*)malloc((size_t)batch * sizeof(float)) : NULL;
  pcv_ctx* ctx = NULL;
  pcv_ingest* ingest = NULL;
  pcv_octree* tree = NULL;
  int rc = pcv_ctx_create(0, NULL, &ctx);
  if (rc != PCV_OK) {
    fprintf(stderr, "no HIP device (pcv_ctx_create: %d); there is no CPU fallback\n", rc);
    return 1;
  }
  rc = pcv_ingest_begin(ctx, n /* NumberOfPoints::num_points() */, with_intensity, &ingest);
  for (unsigned long long at = 0; rc == PCV_OK && at < n; at += batch) {
    const size_t m = (size_t)(n - at < batch ? n - at : batch);
    if (fread(xyz, 3 * sizeof(double), m, fx) != m || fread(rgb, 3, m, fc) != m || (fi && fread(inten, sizeof(float), m, fi) != m)) {
      fprintf(stderr, "short read at point %llu\n", at);
      pcv_ingest_abort(ingest);
      ingest = NULL;
      rc = PCV_E_IO;
      break;
    }
    rc = pcv_ingest_append(ingest, xyz, rgb, inten, m); /* returns when the batch's DMA and kernel are queued */
  }
  if (rc == PCV_OK) {
    pcv_build_params params;
    memset(&params, 0, sizeof(params));
    params.resolution = resolution;
    params.flags = PCV_BUILD_COMPUTE_BBOX; /* the box folded during the ingest (find_bounding_box, generation.rs:256-270) */
    rc = pcv_ingest_finish(ingest, &params, &tree); /* consumes the ingest */
    ingest = NULL;
  } else if (ingest) {
    pcv_ingest_abort(ingest);
  }
  if (rc == PCV_OK) rc = pcv_octree_write_dir(tree, argv[6]);
  if (rc != PCV_OK) fprintf(stderr, "build failed (%d): %s\n", rc, pcv_last_error(ctx));
  else
    printf("%llu points in batches of %llu -> %llu nodes in %s\n", (unsigned long long)pcv_octree_num_points(tree), batch,
           (unsigned long long)pcv_octree_num_nodes(tree), argv[6]);
  pcv_octree_free(tree);
  pcv_ctx_destroy(ctx);
  free(xyz);
  free(rgb);
  free(inten);
  fclose(fx);
  fclose(fc);
  if (fi) fclose(fi);
  return rc == PCV_OK ? 0 : 1;
}
