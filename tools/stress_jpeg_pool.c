// Stress of the JPEG decode pool (comic_jpeg_pool_*) for the CPU sanitizers: twelve batches queued at once over six threads,
// a third of them with a staging buffer too small for all their images, waits polled in reverse order, files that are
// missing / truncated / progressive among the arguments; forty rounds.
//   gcc -O1 -g -fsanitize=address,undefined -pthread -o /tmp/stress_jpeg_pool tools/stress_jpeg_pool.c \
//       comic-compact-image-captioning-with-attention_amd/csrc/jpeg_entropy.c && /tmp/stress_jpeg_pool a.jpg b.jpg missing.jpg ...
// Round 4: clean under ASan + UBSan (ThreadSanitizer does not start in this container).
#include "../include/comic_jpeg.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char** argv) {
  // argv[1..]: files (some may be missing / corrupt)
  const int n = argc - 1;
  comic_jpeg_pool* pool = comic_jpeg_pool_create(6);
  if (getenv("CACHE")) comic_jpeg_pool_enable_cache(pool, 3 << 20);      // (fills up half way through: hits, inserts and refusals mix)
  enum { B = 12 };
  comic_jpeg_info* infos[B]; int32_t* status[B]; int16_t* coef[B]; void* h[B];
  const int64_t cap = 4000000;
  long ok = 0, bad = 0;
  for (int round = 0; round < 40; ++round) {
    for (int b = 0; b < B; ++b) {
      infos[b] = calloc(n, sizeof(comic_jpeg_info)); status[b] = calloc(n, sizeof(int32_t)); coef[b] = malloc(cap * 2);
      const char* paths[64];
      for (int i = 0; i < n; ++i) paths[i] = argv[1 + (i + b + round) % n];
      // even batches dense, odd ones packed (the loader's form); every third one too small for all of its images
      h[b] = (b & 1) ? comic_jpeg_pool_submit_packed(pool, paths, n, infos[b], status[b], (uint16_t*)coef[b], (b % 3 == 2) ? 60000 : cap)
                     : comic_jpeg_pool_submit(pool, paths, n, infos[b], status[b], coef[b], (b % 3 == 2) ? 200000 : cap);
      if (!h[b]) { printf("submit failed\n"); return 1; }
    }
    for (int b = B - 1; b >= 0; --b) {        // waits in reverse order
      int64_t used = -1, px = -1;
      while (comic_jpeg_pool_wait(pool, h[b], 0.001, &used, &px) == 1) {}
      for (int i = 0; i < n; ++i) { if (status[b][i] == 0) ++ok; else ++bad; }
      free(infos[b]); free(status[b]); free(coef[b]);
    }
  }
  // a stage that closes with batches in flight: two batches queued, one waited for with a timeout too short to matter, none
  // released -- destroy finishes the work and frees both handles (LeakSanitizer checks that), the buffers go afterwards
  {
    const char* paths[64];
    for (int i = 0; i < n; ++i) paths[i] = argv[1 + i];
    for (int b = 0; b < 2; ++b) {
      infos[b] = calloc(n, sizeof(comic_jpeg_info)); status[b] = calloc(n, sizeof(int32_t)); coef[b] = malloc(cap * 2);
      h[b] = comic_jpeg_pool_submit_packed(pool, paths, n, infos[b], status[b], (uint16_t*)coef[b], cap);
    }
    int64_t used = 0, px = 0;
    (void)comic_jpeg_pool_wait(pool, h[1], 0.0, &used, &px);
  }
  comic_jpeg_pool_destroy(pool);
  for (int b = 0; b < 2; ++b) { free(infos[b]); free(status[b]); free(coef[b]); }
  printf("images ok %ld other %ld\n", ok, bad);
  return 0;
}
