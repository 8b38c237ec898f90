// Mutation fuzzer of the host half of the split JPEG decoder (csrc/jpeg_entropy.c), for the CPU sanitizers:
//   gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -pthread -o /tmp/fuzz_jpeg tools/fuzz_jpeg.c \
//       comic-compact-image-captioning-with-attention_amd/csrc/jpeg_entropy.c && /tmp/fuzz_jpeg a.jpg b.jpg ...
// 3000 mutations per seed file (truncations, header / scan byte flips, runs of 0xFF), every buffer allocated at its exact size;
// every case also through the packed decode of a pool (same verdict as the dense decode, or the blob too small).
// Round 4: 12 000 runs over four seed files (4:2:0, 4:2:2 + restart markers, 4:4:4 optimised tables, greyscale): clean.
#include "../include/comic_jpeg.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static unsigned long long s = 88172645463325252ull;
static unsigned rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
int main(int argc, char** argv) {
  long total = 0, ok = 0, unsup = 0, corrupt = 0, pk_ok = 0;
  comic_jpeg_pool* pool = comic_jpeg_pool_create(2);        // the packed decode (the loader's form) goes through the pool
  const long pk_cap = 600000;
  unsigned short* pk = malloc(pk_cap * 2);
  for (int a = 1; a < argc; ++a) {
    FILE* f = fopen(argv[a], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); rewind(f);
    unsigned char* base = malloc(n); fread(base, 1, n, f); fclose(f);
    for (int it = 0; it < 3000; ++it) {
      long m = n;
      unsigned char* d = malloc(n);           // exact size: reads past the end are caught
      memcpy(d, base, n);
      int kind = rnd() % 6;
      if (kind == 0) m = 1 + rnd() % n;                                   // truncation
      else if (kind == 5) { long at = rnd() % n, len = rnd() % 64; for (long k = at; k < at + len && k < n; ++k) d[k] = 0xFF; }
      else { int flips = 1 + rnd() % (kind * 4); for (int k = 0; k < flips; ++k) { long at = (kind == 1) ? rnd() % (n < 700 ? n : 700) : rnd() % n; d[at] = (unsigned char)rnd(); } }
      unsigned char* e = malloc(m); memcpy(e, d, m); free(d);
      comic_jpeg_info info;
      int rc = comic_jpeg_read_header(e, m, &info);
      if (rc == 0 && info.coef_count > 0 && info.coef_count < 40000000) {
        short* coef = malloc(info.coef_count * 2);
        rc = comic_jpeg_decode_coefficients(e, m, &info, coef);
        free(coef);
      }
      ++total; if (rc == 0) ++ok; else if (rc == 1) ++unsup; else ++corrupt;
      {
        FILE* t = fopen("/tmp/fuzz_jpeg_case.jpg", "wb"); fwrite(e, 1, m, t); fclose(t);
        const char* one[1] = {"/tmp/fuzz_jpeg_case.jpg"};
        comic_jpeg_info pi; int st = 99; long long used = 0, planes = 0;
        void* h = comic_jpeg_pool_submit_packed(pool, one, 1, &pi, &st, pk, pk_cap);
        while (comic_jpeg_pool_wait(pool, h, 1.0, (int64_t*)&used, (int64_t*)&planes) == 1) {}
        if (st == 0) ++pk_ok;
        if ((st == 0) != (rc == 0) && !(rc == 0 && st == COMIC_JPEG_TOO_SMALL)) { printf("dense rc %d, packed status %d\n", rc, st); return 1; }
      }
      free(e);
    }
    free(base);
  }
  comic_jpeg_pool_destroy(pool);
  free(pk);
  printf("runs %ld ok %ld unsupported %ld corrupt %ld; packed ok %ld\n", total, ok, unsup, corrupt, pk_ok);
  return 0;
}
