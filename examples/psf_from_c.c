/* The inner drop-in boundary called from plain C -- no Python, no torch, nothing but include/wayne_hip.h and
 * libwayne_hip.so: what a maintainer's cgo / JNI / ctypes stub reduces to.
 *
 *   gcc -std=c99 -I include examples/psf_from_c.c -o psf_from_c -L wayne_amd -lwayne_hip -Wl,-rpath,$PWD/wayne_amd
 *   ./psf_from_c in.bin out.bin
 *
 * in.bin : int32 size, nr, nc, test, threads; then counts[size] (int32), x, y, ratio, sigma_l, sigma_h (double[size])
 *          -- the arguments of the reference's PSF() (wayne/pyparallel_menu.h:1-3)
 * out.bin: the frame, int32[nr * nc], thrown in the replay mode: equal to the reference's bit for bit.
 * tests/test_psf_gpu.py builds and runs it on a golden vector of the reference. */
#include <stdio.h>
#include <stdlib.h>

#include "wayne_hip.h"

static void *read_block(FILE *f, size_t n, size_t width) {
  void *p = malloc(n > 0 ? n * width : 1);
  if (!p || fread(p, width, n, f) != n) { fprintf(stderr, "short input\n"); exit(2); }
  return p;
}

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int32_t head[5];
  if (fread(head, sizeof head[0], 5, f) != 5) { fprintf(stderr, "short header\n"); return 2; }
  const int size = head[0], nr = head[1], nc = head[2], test = head[3], threads = head[4];
  int32_t *counts = read_block(f, (size_t)size, sizeof(int32_t));
  double *x = read_block(f, (size_t)size, sizeof(double)), *y = read_block(f, (size_t)size, sizeof(double));
  double *ratio = read_block(f, (size_t)size, sizeof(double));
  double *sl = read_block(f, (size_t)size, sizeof(double)), *sh = read_block(f, (size_t)size, sizeof(double));
  fclose(f);

  int status = 0;
  wayne_ctx *ctx = wayne_ctx_create(0, &status);
  if (!ctx) { fprintf(stderr, "wayne_ctx_create: %s\n", wayne_strerror(status)); return 1; }   /* no GPU: fails loudly */
  int32_t *frame = malloc((size_t)nr * nc * sizeof(int32_t));
  int rc = wayne_psf_apply(ctx, counts, size, x, y, ratio, sl, sh, nr, nc, (uint32_t)test, threads, WAYNE_RNG_REPLAY,
                           0u, 0u, frame);
  if (rc != WAYNE_OK) { fprintf(stderr, "wayne_psf_apply: %s\n", wayne_last_error(ctx)); return 1; }
  f = fopen(argv[2], "wb");
  if (!f || fwrite(frame, sizeof(int32_t), (size_t)nr * nc, f) != (size_t)nr * nc) { perror(argv[2]); return 2; }
  fclose(f);
  long long total = 0;
  for (long i = 0; i < (long)nr * nc; ++i) total += frame[i];
  printf("abi %d: %d bins -> %d x %d frame, %lld electrons on it\n", wayne_abi_version(), size, nr, nc, total);
  wayne_ctx_destroy(ctx);
  free(frame); free(counts); free(x); free(y); free(ratio); free(sl); free(sh);
  return 0;
}
