// Host half of the split JPEG decoder (include/comic_jpeg.h): marker parsing + Huffman decoding of a baseline
// (SOF0 / SOF1, 8-bit, one interleaved scan) JPEG into quantised DCT coefficients.  Everything after the entropy coding
// -- dequantisation, inverse DCT, upsampling, colour conversion -- runs on the device (csrc/jpeg_pixels.hip).
//
// Stands where the reference's tf.data map calls tf.image.decode_jpeg (libjpeg) per image:
// common/inputs/manager_image_caption.py:163-175 -> preprocessing/inception_preprocessing_radix.py.  The bit stream
// format is ITU-T T.81 (Annex F.2.2 decoding procedures, Annex C code-table generation); nothing here comes from
// libjpeg's sources.  Plain C, no GPU runtime: loader threads call it through ctypes with the interpreter lock released.
#include "../../include/comic_jpeg.h"

#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#define LOOK 10                     // bits of the one-probe code tables

typedef struct {
  // one probe of LOOK bits: (code length << 8) | symbol, 0 when the code is longer than LOOK bits
  uint16_t look[1 << LOOK];
  // AC tables only: a whole (run, size, value) triple that fits the probe: (value << 8) | (run << 4) | total bits; 0: none
  int16_t fast_ac[1 << LOOK];
  // the long way (T.81 F.2.2.3): per code length the largest code, and the index of its first symbol
  int32_t maxcode[18];
  int32_t valoff[17];
  uint8_t vals[256];
  int present;
} HuffTable;

typedef struct {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t bits;                    // the next bits of the stream, most significant first
  int nbits;
  int marker;                       // a marker was reached: zeros are fed from here on
  int fill;                         // zero bits fed behind a marker / the end of the data (a complete stream consumes none)
} BitReader;

static const uint8_t kZigzag[64 + 16] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    // a corrupt run may step past 63: the overflow lands on the last coefficient instead of outside the block
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

// Refill to >= 56 valid bits.  Fast way: eight stream bytes at once while none of them is 0xFF (stuffing and markers take
// the byte-wise way).  Only whole bytes are counted in nbits; the bits of the next, partly loaded byte already sit below them
// -- the same bits the next refill ORs in again from the same address, so they need no masking.
#define BR_REFILL(bits, nbits, p, end, br)                                                     \
  do {                                                                                         \
    uint64_t v__;                                                                              \
    int fast__ = 0;                                                                            \
    if ((end) - (p) >= 8) {                                                                    \
      memcpy(&v__, (p), 8);                                                                    \
      const uint64_t inv__ = ~v__;                                                             \
      fast__ = !((inv__ - 0x0101010101010101ull) & ~inv__ & 0x8080808080808080ull);            \
    }                                                                                          \
    if (fast__) {                                                                              \
      (bits) |= __builtin_bswap64(v__) >> (nbits);                                             \
      (p) += (63 - (nbits)) >> 3;                                                              \
      (nbits) |= 56;                                                                           \
    } else {                                                                                   \
      (br)->p = (p); (br)->bits = (bits); (br)->nbits = (nbits);                               \
      br_refill_slow(br);                                                                      \
      (p) = (br)->p; (bits) = (br)->bits; (nbits) = (br)->nbits;                               \
    }                                                                                          \
  } while (0)

static void br_refill_slow(BitReader* br) {
  // (bits below the counted ones may hold the head of the byte at p: it is ORed in again, unchanged)
  while (br->nbits <= 56) {
    unsigned b = 0;
    if (br->marker || br->p >= br->end) br->fill += 8;
    else {
      b = *br->p;
      if (b == 0xFF) {
        if (br->p + 1 < br->end && br->p[1] == 0x00) br->p += 2;          // a stuffed zero after a data 0xFF
        else {
          br->marker = 1;                                                  // p stays on the marker's 0xFF
          br->fill += 8;
          b = 0;
        }
      } else {
        br->p += 1;
      }
    }
    br->bits |= (uint64_t)b << (56 - br->nbits);
    br->nbits += 8;
  }
}

#define PEEK(bits, n) ((unsigned)((bits) >> (64 - (n))))
#define SKIP(bits, nbits, n) \
  do {                       \
    (bits) <<= (n);          \
    (nbits) -= (n);          \
  } while (0)
// the value of an `s`-bit magnitude field (T.81 F.2.2.1, EXTEND): codes with a leading 0 bit are negative
static inline int extend(unsigned v, int s) { return (int)v - (int)(((v >> (s - 1)) ^ 1u) * ((1u << s) - 1u)); }

static int build_table(HuffTable* t, const uint8_t counts[16], const uint8_t* vals, int nvals, int is_ac) {
  memset(t, 0, sizeof(*t));
  memcpy(t->vals, vals, (size_t)nvals);
  int code = 0, k = 0;
  for (int len = 1; len <= 16; ++len) {
    t->valoff[len] = k - code;
    for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
      if (code >= (1 << len)) return COMIC_JPEG_CORRUPT;
      if (len <= LOOK) {
        const int first = code << (LOOK - len), span = 1 << (LOOK - len);
        for (int j = 0; j < span; ++j) t->look[first + j] = (uint16_t)((len << 8) | vals[k]);
      }
    }
    t->maxcode[len] = counts[len - 1] ? code - 1 : -1;
    code <<= 1;
  }
  t->maxcode[17] = 0x7fffffff;
  if (is_ac) {
    for (int i = 0; i < (1 << LOOK); ++i) {
      const unsigned e = t->look[i];
      if (!e) continue;
      const int len = (int)(e >> 8), rs = (int)(e & 255), run = rs >> 4, size = rs & 15;
      if (size == 0 || len + size > LOOK) continue;
      const unsigned field = ((unsigned)i >> (LOOK - len - size)) & ((1u << size) - 1u);
      const int value = (int)field - (int)(((field >> (size - 1)) ^ 1u) * ((1u << size) - 1u));
      if (value >= -128 && value <= 127) t->fast_ac[i] = (int16_t)(value * 256 + run * 16 + len + size);
    }
  }
  t->present = 1;
  return COMIC_JPEG_OK;
}

// the long way of a code that does not fit the probe (T.81 F.2.2.3); returns (length << 8) | symbol, -1 for no such code
static int decode_long(uint64_t bits, const HuffTable* t) {
  const int code16 = (int)PEEK(bits, 16);
  int len = LOOK + 1;
  while (len <= 16 && (code16 >> (16 - len)) > t->maxcode[len]) ++len;
  if (len > 16) return -1;
  const int idx = (code16 >> (16 - len)) + t->valoff[len];
  return idx >= 0 && idx < 256 ? (len << 8) | t->vals[idx] : -1;
}

static inline int decode_block(BitReader* br, const HuffTable* dc, const HuffTable* ac, int* pred, int16_t* blk) {
  uint64_t bits = br->bits;
  int nbits = br->nbits;
  const uint8_t* p = br->p;
  const uint8_t* const end = br->end;
  memset(blk, 0, 64 * sizeof(int16_t));                // (block by block: the lines are still in L1 when the coefficients land)
  BR_REFILL(bits, nbits, p, end, br);                  // >= 56 bits: the DC code (<= 16) and its field (<= 11 valid)
  {
    int e = dc->look[PEEK(bits, LOOK)];
    if (!e) e = decode_long(bits, dc);
    if (e < 0) return COMIC_JPEG_CORRUPT;
    SKIP(bits, nbits, e >> 8);
    const int t = e & 255;
    if (t > 15) return COMIC_JPEG_CORRUPT;
    if (t) {
      *pred += extend(PEEK(bits, t), t);
      SKIP(bits, nbits, t);
    }
    blk[0] = (int16_t)*pred;
  }
  int k = 1;
  do {
    if (nbits < 32) BR_REFILL(bits, nbits, p, end, br);   // a code (<= 16) and its field (<= 15) per round
    const unsigned probe = PEEK(bits, LOOK);
    const int f = ac->fast_ac[probe];
    if (f) {
      k += (f >> 4) & 15;
      SKIP(bits, nbits, f & 15);
      blk[kZigzag[k++]] = (int16_t)(f >> 8);
      continue;
    }
    int e = ac->look[probe];
    if (!e) e = decode_long(bits, ac);
    if (e < 0) return COMIC_JPEG_CORRUPT;
    SKIP(bits, nbits, e >> 8);
    const int s = e & 15, r = (e >> 4) & 15;
    if (s == 0) {
      if (r != 15) break;                              // end of block
      k += 16;
    } else {
      k += r;
      blk[kZigzag[k++]] = (int16_t)extend(PEEK(bits, s), s);
      SKIP(bits, nbits, s);
    }
  } while (k < 64);
  br->bits = bits;
  br->nbits = nbits;
  br->p = p;
  return COMIC_JPEG_OK;
}

// The same block in the packed form of the loader's batches (comic_jpeg_pool_submit_packed): the DC coefficient goes to *dcv, every
// non-zero AC coefficient becomes a 16-bit entry (natural position << 10) | (value & 1023), or -- for |value| >= 512, rare --
// the pair (position, value as int16); no 128-byte clear and no scattered 2-byte stores per block, 3-4x fewer bytes for the bus.
// Returns the number of 16-bit entries (<= 126), < 0 on a bad stream.
static inline int decode_block_packed(BitReader* br, const HuffTable* dc, const HuffTable* ac, int* pred, int16_t* dcv, uint16_t* ent) {
  uint64_t bits = br->bits;
  int nbits = br->nbits;
  const uint8_t* p = br->p;
  const uint8_t* const end = br->end;
  int cnt = 0;
  BR_REFILL(bits, nbits, p, end, br);
  {
    int e = dc->look[PEEK(bits, LOOK)];
    if (!e) e = decode_long(bits, dc);
    if (e < 0) return COMIC_JPEG_CORRUPT;
    SKIP(bits, nbits, e >> 8);
    const int t = e & 255;
    if (t > 15) return COMIC_JPEG_CORRUPT;
    if (t) {
      *pred += extend(PEEK(bits, t), t);
      SKIP(bits, nbits, t);
    }
    *dcv = (int16_t)*pred;
  }
  int k = 1;
  do {
    if (nbits < 32) BR_REFILL(bits, nbits, p, end, br);
    const unsigned probe = PEEK(bits, LOOK);
    const int f = ac->fast_ac[probe];
    if (f) {                                             // (values of this table are within -128 .. 127)
      k += (f >> 4) & 15;
      SKIP(bits, nbits, f & 15);
      ent[cnt++] = (uint16_t)((kZigzag[k++] << 10) | ((f >> 8) & 1023));
      continue;
    }
    int e = ac->look[probe];
    if (!e) e = decode_long(bits, ac);
    if (e < 0) return COMIC_JPEG_CORRUPT;
    SKIP(bits, nbits, e >> 8);
    const int s = e & 15, r = (e >> 4) & 15;
    if (s == 0) {
      if (r != 15) break;
      k += 16;
    } else {
      k += r;
      const int v = extend(PEEK(bits, s), s);
      SKIP(bits, nbits, s);
      const int pos = kZigzag[k++];
      if (v >= -512 && v <= 511 && pos != 0) ent[cnt++] = (uint16_t)((pos << 10) | (v & 1023));
      else {
        ent[cnt++] = (uint16_t)pos;
        ent[cnt++] = (uint16_t)(int16_t)v;
      }
    }
  } while (k < 64);
  br->bits = bits;
  br->nbits = nbits;
  br->p = p;
  return cnt;
}

// ---- markers -------------------------------------------------------------------------------------------------------------
typedef struct {
  comic_jpeg_info info;
  HuffTable dc[4], ac[4];
  uint16_t qt[4][64];
  int qt_present[4];
  int comp_id[3], comp_tq[3], comp_td[3], comp_ta[3], comp_h[3], comp_v[3];
  int saw_jfif, adobe_transform;   // adobe_transform -1: no APP14
  const uint8_t* scan;              // first byte of entropy-coded data
} Parsed;

static inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

static int parse(const uint8_t* data, int64_t n, Parsed* ps, int want_tables) {
  memset(&ps->info, 0, sizeof(ps->info));
  memset(ps->qt_present, 0, sizeof(ps->qt_present));
  if (want_tables)
    for (int i = 0; i < 4; ++i) ps->dc[i].present = ps->ac[i].present = 0;
  ps->saw_jfif = 0;
  ps->adobe_transform = -1;
  ps->scan = NULL;
  if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) return COMIC_JPEG_CORRUPT;
  const uint8_t* p = data + 2;
  const uint8_t* end = data + n;
  int have_frame = 0;
  comic_jpeg_info* in = &ps->info;
  for (;;) {
    while (p < end && *p != 0xFF) ++p;                 // (garbage between segments is skipped, as decoders do)
    while (p < end && *p == 0xFF) ++p;                 // fill bytes
    if (p >= end) return COMIC_JPEG_CORRUPT;
    const int m = *p++;
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;      // no payload
    if (m == 0xD9) return COMIC_JPEG_CORRUPT;                               // EOI before a scan
    if (end - p < 2) return COMIC_JPEG_CORRUPT;
    const int len = be16(p);
    if (len < 2 || end - p < len) return COMIC_JPEG_CORRUPT;
    const uint8_t* s = p + 2;
    const uint8_t* se = p + len;
    p = se;
    if (m == 0xC0 || m == 0xC1) {                                           // baseline / extended sequential, Huffman
      if (have_frame || se - s < 6) return COMIC_JPEG_CORRUPT;
      if (s[0] != 8) return COMIC_JPEG_UNSUPPORTED;                         // 12-bit samples
      in->height = be16(s + 1);
      in->width = be16(s + 3);
      in->ncomp = s[5];
      if (in->height == 0 || in->width == 0) return COMIC_JPEG_UNSUPPORTED; // (DNL-defined height)
      if (in->ncomp != 1 && in->ncomp != 3) return COMIC_JPEG_UNSUPPORTED;  // CMYK / YCCK
      if (se - s < 6 + 3 * in->ncomp) return COMIC_JPEG_CORRUPT;
      for (int c = 0; c < in->ncomp; ++c) {
        ps->comp_id[c] = s[6 + 3 * c];
        ps->comp_h[c] = s[7 + 3 * c] >> 4;
        ps->comp_v[c] = s[7 + 3 * c] & 15;
        ps->comp_tq[c] = s[8 + 3 * c];
        if (ps->comp_tq[c] > 3) return COMIC_JPEG_CORRUPT;
      }
      have_frame = 1;
    } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
      return COMIC_JPEG_UNSUPPORTED;                                        // progressive, lossless, arithmetic ...
    } else if (m == 0xCC) {
      return COMIC_JPEG_UNSUPPORTED;                                        // arithmetic conditioning
    } else if (m == 0xC4) {                                                 // DHT: any number of tables
      while (s < se) {
        if (se - s < 17) return COMIC_JPEG_CORRUPT;
        const int tc = s[0] >> 4, th = s[0] & 15;
        if (tc > 1 || th > 3) return COMIC_JPEG_CORRUPT;
        int total = 0;
        for (int i = 0; i < 16; ++i) total += s[1 + i];
        if (total > 256 || se - s < 17 + total) return COMIC_JPEG_CORRUPT;
        if (want_tables) {
          const int rc = build_table(tc ? &ps->ac[th] : &ps->dc[th], s + 1, s + 17, total, tc);
          if (rc) return rc;
        }
        s += 17 + total;
      }
    } else if (m == 0xDB) {                                                 // DQT
      while (s < se) {
        const int pq = s[0] >> 4, tq = s[0] & 15;
        if (pq > 1 || tq > 3 || se - s < 1 + 64 * (pq + 1)) return COMIC_JPEG_CORRUPT;
        for (int i = 0; i < 64; ++i) ps->qt[tq][kZigzag[i]] = (uint16_t)(pq ? be16(s + 1 + 2 * i) : s[1 + i]);
        ps->qt_present[tq] = 1;
        s += 1 + 64 * (pq + 1);
      }
    } else if (m == 0xDD) {                                                 // DRI
      if (se - s < 2) return COMIC_JPEG_CORRUPT;
      in->restart_interval = be16(s);
    } else if (m == 0xE0) {
      if (se - s >= 5 && !memcmp(s, "JFIF", 5)) ps->saw_jfif = 1;
    } else if (m == 0xEE) {
      if (se - s >= 12 && !memcmp(s, "Adobe", 5)) ps->adobe_transform = s[11];
    } else if (m == 0xDA) {                                                 // SOS
      if (!have_frame || se - s < 1) return COMIC_JPEG_CORRUPT;
      const int ns = s[0];
      if (ns != in->ncomp) return COMIC_JPEG_UNSUPPORTED;                   // one scan per component
      if (se - s < 1 + 2 * ns + 3) return COMIC_JPEG_CORRUPT;
      for (int c = 0; c < ns; ++c) {
        if (s[1 + 2 * c] != ps->comp_id[c]) return COMIC_JPEG_UNSUPPORTED;  // components out of frame order
        ps->comp_td[c] = s[2 + 2 * c] >> 4;
        ps->comp_ta[c] = s[2 + 2 * c] & 15;
        if (ps->comp_td[c] > 3 || ps->comp_ta[c] > 3) return COMIC_JPEG_CORRUPT;
      }
      if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return COMIC_JPEG_UNSUPPORTED;
      ps->scan = se;
      break;
    }
  }
  // colour space as libjpeg decides it for three components: JFIF says YCbCr; else Adobe's transform flag; else the ids
  if (in->ncomp == 3) {
    if (!ps->saw_jfif) {
      if (ps->adobe_transform == 0) return COMIC_JPEG_UNSUPPORTED;                                     // RGB
      if (ps->adobe_transform < 0 && ps->comp_id[0] == 'R' && ps->comp_id[1] == 'G' && ps->comp_id[2] == 'B')
        return COMIC_JPEG_UNSUPPORTED;
      if (ps->adobe_transform > 1) return COMIC_JPEG_UNSUPPORTED;
    }
    if (ps->comp_h[1] != 1 || ps->comp_v[1] != 1 || ps->comp_h[2] != 1 || ps->comp_v[2] != 1) return COMIC_JPEG_UNSUPPORTED;
    const int h = ps->comp_h[0], v = ps->comp_v[0];
    if (!((h == 1 && v == 1) || (h == 2 && v == 1) || (h == 2 && v == 2))) return COMIC_JPEG_UNSUPPORTED;
    in->hmax = h;
    in->vmax = v;
  } else {
    // a single component is never interleaved: its MCU is one block whatever the factors say (T.81 A.2.2)
    ps->comp_h[0] = ps->comp_v[0] = 1;
    in->hmax = in->vmax = 1;
  }
  in->mcus_x = (in->width + 8 * in->hmax - 1) / (8 * in->hmax);
  in->mcus_y = (in->height + 8 * in->vmax - 1) / (8 * in->vmax);
  int64_t off = 0;
  for (int c = 0; c < in->ncomp; ++c) {
    if (!ps->qt_present[ps->comp_tq[c]]) return COMIC_JPEG_CORRUPT;
    memcpy(in->quant[c], ps->qt[ps->comp_tq[c]], sizeof(in->quant[c]));
    in->blocks_w[c] = in->mcus_x * ps->comp_h[c];
    in->blocks_h[c] = in->mcus_y * ps->comp_v[c];
    in->comp_w[c] = (in->width * ps->comp_h[c] + in->hmax - 1) / in->hmax;
    in->comp_h[c] = (in->height * ps->comp_v[c] + in->vmax - 1) / in->vmax;
    in->coef_off[c] = off;
    off += (int64_t)in->blocks_w[c] * in->blocks_h[c] * 64;
  }
  in->coef_count = off;
  // fancy upsampling needs more than two chroma columns (jdsample.c picks the box filter below that): leave those to PIL
  if (in->ncomp == 3 && in->hmax == 2 && in->comp_w[1] <= 2) return COMIC_JPEG_UNSUPPORTED;
  if (want_tables)
    for (int c = 0; c < in->ncomp; ++c)
      if (!ps->dc[ps->comp_td[c]].present || !ps->ac[ps->comp_ta[c]].present) return COMIC_JPEG_CORRUPT;
  return COMIC_JPEG_OK;
}

int comic_jpeg_read_header(const uint8_t* data, int64_t n, comic_jpeg_info* info) {
  if (!data || !info) return COMIC_JPEG_CORRUPT;
  Parsed* ps = (Parsed*)malloc(sizeof(Parsed));
  if (!ps) return COMIC_JPEG_IO;
  const int rc = parse(data, n, ps, 0);
  *info = ps->info;
  free(ps);
  return rc;
}

// (cloned for BMI2: the bit reader is variable shifts all the way down)
__attribute__((target_clones("default", "bmi2")))
static int decode_scan(const uint8_t* data, int64_t n, Parsed* ps, int16_t* coef) {
  const comic_jpeg_info* in = &ps->info;
  BitReader br = {ps->scan, data + n, 0, 0, 0, 0};
  int pred[3] = {0, 0, 0};
  int to_restart = in->restart_interval;
  int next_rst = 0;
  const int nc = in->ncomp;
  for (int my = 0; my < in->mcus_y; ++my) {
    for (int mx = 0; mx < in->mcus_x; ++mx) {
      if (in->restart_interval && to_restart == 0) {
        // byte-align, take the RSTn marker, reset the predictions (T.81 F.2.2.4 / E.2.4)
        if (br.fill > br.nbits) return COMIC_JPEG_CORRUPT;                 // the interval ended inside an MCU
        const uint8_t* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
          if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) return COMIC_JPEG_CORRUPT;   // another marker
          ++q;
        }
        if (q + 1 >= br.end || q - br.p > 16) return COMIC_JPEG_CORRUPT;
        if ((q[1] & 7) != next_rst) return COMIC_JPEG_CORRUPT;
        next_rst = (next_rst + 1) & 7;
        br.p = q + 2;
        br.bits = 0;
        br.nbits = 0;
        br.marker = 0;
        br.fill = 0;
        pred[0] = pred[1] = pred[2] = 0;
        to_restart = in->restart_interval;
      }
      for (int c = 0; c < nc; ++c) {
        const HuffTable* dc = &ps->dc[ps->comp_td[c]];
        const HuffTable* ac = &ps->ac[ps->comp_ta[c]];
        const int hs = ps->comp_h[c], vs = ps->comp_v[c];
        for (int v = 0; v < vs; ++v)
          for (int h = 0; h < hs; ++h) {
            int16_t* blk = coef + in->coef_off[c] + ((int64_t)(my * vs + v) * in->blocks_w[c] + (mx * hs + h)) * 64;
            const int rc = decode_block(&br, dc, ac, &pred[c], blk);
            if (rc) return rc;
          }
      }
      --to_restart;
    }
  }
  // a truncated scan decodes the zeros fed behind its end: PIL refuses such a file, so does this decoder
  return br.fill > br.nbits ? COMIC_JPEG_CORRUPT : COMIC_JPEG_OK;
}

__attribute__((target_clones("default", "bmi2")))
// Packed image, in 16-bit units: [desc: blocks x uint32][dc: blocks x int16][entries].  desc[g] = (first entry << 7) | entries
// of block g (plane order: coef_off[c] / 64 + row * blocks_w + column); the entries lie in decoding (MCU) order; *total = their
// number.  `ent` must hold 126 entries per block.
static int decode_scan_packed(const uint8_t* data, int64_t n, Parsed* ps, uint32_t* desc, int16_t* dcs, uint16_t* ent, int64_t* total) {
  int64_t cur = 0;
  const comic_jpeg_info* in = &ps->info;
  BitReader br = {ps->scan, data + n, 0, 0, 0, 0};
  int pred[3] = {0, 0, 0};
  int to_restart = in->restart_interval;
  int next_rst = 0;
  const int nc = in->ncomp;
  for (int my = 0; my < in->mcus_y; ++my) {
    for (int mx = 0; mx < in->mcus_x; ++mx) {
      if (in->restart_interval && to_restart == 0) {
        // byte-align, take the RSTn marker, reset the predictions (T.81 F.2.2.4 / E.2.4)
        if (br.fill > br.nbits) return COMIC_JPEG_CORRUPT;                 // the interval ended inside an MCU
        const uint8_t* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
          if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) return COMIC_JPEG_CORRUPT;   // another marker
          ++q;
        }
        if (q + 1 >= br.end || q - br.p > 16) return COMIC_JPEG_CORRUPT;
        if ((q[1] & 7) != next_rst) return COMIC_JPEG_CORRUPT;
        next_rst = (next_rst + 1) & 7;
        br.p = q + 2;
        br.bits = 0;
        br.nbits = 0;
        br.marker = 0;
        br.fill = 0;
        pred[0] = pred[1] = pred[2] = 0;
        to_restart = in->restart_interval;
      }
      for (int c = 0; c < nc; ++c) {
        const HuffTable* dc = &ps->dc[ps->comp_td[c]];
        const HuffTable* ac = &ps->ac[ps->comp_ta[c]];
        const int hs = ps->comp_h[c], vs = ps->comp_v[c];
        for (int v = 0; v < vs; ++v)
          for (int h = 0; h < hs; ++h) {
            const int64_t g = in->coef_off[c] / 64 + (int64_t)(my * vs + v) * in->blocks_w[c] + (mx * hs + h);
            const int cnt = decode_block_packed(&br, dc, ac, &pred[c], dcs + g, ent + cur);
            if (cnt < 0) return cnt;
            desc[g] = (uint32_t)(cur << 7) | (uint32_t)cnt;
            cur += cnt;
          }
      }
      --to_restart;
    }
  }
  // a truncated scan decodes the zeros fed behind its end: PIL refuses such a file, so does this decoder
  *total = cur;
  return br.fill > br.nbits ? COMIC_JPEG_CORRUPT : COMIC_JPEG_OK;
}


int comic_jpeg_decode_coefficients(const uint8_t* data, int64_t n, const comic_jpeg_info* info, int16_t* coef) {
  if (!data || !info || !coef) return COMIC_JPEG_CORRUPT;
  Parsed* ps = (Parsed*)malloc(sizeof(Parsed));
  if (!ps) return COMIC_JPEG_IO;
  int rc = parse(data, n, ps, 1);
  if (rc == COMIC_JPEG_OK && (ps->info.coef_count != info->coef_count || ps->info.width != info->width ||
                              ps->info.height != info->height))
    rc = COMIC_JPEG_CORRUPT;
  if (rc == COMIC_JPEG_OK) rc = decode_scan(data, n, ps, coef);
  free(ps);
  return rc;
}

static int read_file(const char* path, uint8_t** out, int64_t* out_n);

int comic_jpeg_decode_file(const char* path, comic_jpeg_info* info, int16_t* coef, int64_t coef_capacity) {
  if (!path || !info) return COMIC_JPEG_CORRUPT;
  uint8_t* data = NULL;
  int64_t n = 0;
  int rc = read_file(path, &data, &n);
  if (rc) return rc;
  Parsed* ps = (Parsed*)malloc(sizeof(Parsed));
  rc = ps ? parse(data, n, ps, 1) : COMIC_JPEG_IO;
  if (ps) *info = ps->info;
  if (rc == COMIC_JPEG_OK) {
    if (!coef || coef_capacity < ps->info.coef_count) rc = COMIC_JPEG_TOO_SMALL;
    else rc = decode_scan(data, n, ps, coef);
  }
  free(ps);
  free(data);
  return rc;
}

// ---- the decode pool ---------------------------------------------------------------------------------------------------------
// Two passes per batch so that the coefficients of its images lie back to back (ONE host-to-device copy per batch; a copy per
// image cost the consumer thread 30 us each): pass 1 reads every file and parses its headers, the thread that finishes the
// last one lays the images out, pass 2 decodes the scans.  Threads that find a batch between its passes go on to the next
// queued batch.
#include <pthread.h>
#include <time.h>

struct CacheEntry;
typedef struct {
  uint8_t* data;                   // the file, kept from pass 1 to pass 2
  int64_t n;
  Parsed* ps;                      // its tables
  const struct CacheEntry* hit;    // the image's coefficients are in the pool's cache: no file read, no Huffman decoding
} Item;

typedef struct Batch {
  char** paths;
  Item* items;
  int n, next1, done1, next2, done2;
  comic_jpeg_info* infos;
  int32_t* status;
  int16_t* coef;
  int64_t capacity, used;
  int packed;                      // comic_jpeg_pool_submit_packed: `coef` is the packed blob (16-bit units), one pass per image
  struct comic_jpeg_pool* pool;
  struct Batch* link;
} Batch;

// ---- coefficient cache (comic_jpeg_pool_enable_cache): epochs after the first skip the entropy decoding ------------------------
// An image is kept in the packed form of the loader's batches -- per 8x8 block a descriptor and the DC value, per non-zero AC
// coefficient a 16-bit entry: 3-4x smaller than the dense int16 blocks -- under its path; entries are never evicted (insertion stops at
// the byte limit), so a hit's pointer stays valid until the pool is destroyed.  Readers and the inserting thread meet under a
// read-write lock.
typedef struct CacheEntry {
  char* path;
  uint64_t hash;
  comic_jpeg_info info;
  uint16_t* packed;                // the packed form of decode_scan_packed
  int64_t n_u16;
  int64_t file_size, mtime_ns;     // of the file when it was decoded: a file that changed since is decoded again
  struct CacheEntry* next;
} CacheEntry;

typedef struct {
  pthread_rwlock_t lock;
  CacheEntry** buckets;
  int64_t n_buckets, max_bytes, bytes, entries, hits;
} CoefCache;

static uint64_t path_hash(const char* s) {
  uint64_t h = 1469598103934665603ull;                  // FNV-1a
  for (; *s; ++s) h = (h ^ (uint8_t)*s) * 1099511628211ull;
  return h;
}

static int file_stamp(const char* path, int64_t* size, int64_t* mtime_ns) {
  struct stat st;
  if (stat(path, &st)) return -1;
  *size = (int64_t)st.st_size;
  *mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
  return 0;
}

static const CacheEntry* cache_lookup(CoefCache* c, const char* path) {
  if (!c || !c->buckets) return NULL;
  const uint64_t h = path_hash(path);
  pthread_rwlock_rdlock(&c->lock);
  const CacheEntry* e = c->buckets[h % (uint64_t)c->n_buckets];
  while (e && !(e->hash == h && !strcmp(e->path, path))) e = e->next;
  pthread_rwlock_unlock(&c->lock);
  if (e) {                         // served only while the file is the one that was decoded (size and modification time)
    int64_t size = 0, mt = 0;
    if (file_stamp(path, &size, &mt) || size != e->file_size || mt != e->mtime_ns) return NULL;
    __atomic_fetch_add(&c->hits, 1, __ATOMIC_RELAXED);
  }
  return e;
}

// dense blocks <-> the packed form of decode_scan_packed, in 16-bit units: [desc 2 x blocks][dc blocks][entries]
static int64_t dense_to_packed(const int16_t* coef, int64_t blocks, uint16_t* out) {     // out == NULL: size only
  uint32_t* desc = (uint32_t*)out;
  int16_t* dcs = out ? (int16_t*)(out + 2 * blocks) : NULL;
  uint16_t* ent = out ? out + 3 * blocks : NULL;
  int64_t cur = 0;
  for (int64_t b = 0; b < blocks; ++b) {
    const int16_t* blk = coef + b * 64;
    const int64_t beg = cur;
    if (out) dcs[b] = blk[0];
    for (int k = 1; k < 64; ++k)
      if (blk[k]) {
        const int v = blk[k];
        if (v >= -512 && v <= 511) {
          if (out) ent[cur] = (uint16_t)((k << 10) | (v & 1023));
          cur += 1;
        } else {
          if (out) {
            ent[cur] = (uint16_t)k;
            ent[cur + 1] = (uint16_t)(int16_t)v;
          }
          cur += 2;
        }
      }
    if (out) desc[b] = (uint32_t)(beg << 7) | (uint32_t)(cur - beg);
  }
  return (3 * blocks + cur + 1) & ~(int64_t)1;           // whole 32-bit words: the next image's descriptors stay aligned
}

static void packed_to_dense(const uint16_t* pk, int64_t blocks, int16_t* coef) {
  const uint32_t* desc = (const uint32_t*)pk;
  const int16_t* dcs = (const int16_t*)(pk + 2 * blocks);
  const uint16_t* ent = pk + 3 * blocks;
  memset(coef, 0, (size_t)blocks * 64 * sizeof(int16_t));
  for (int64_t b = 0; b < blocks; ++b) {
    const uint16_t* e = ent + (desc[b] >> 7);
    const int n = (int)(desc[b] & 127);
    int16_t* blk = coef + b * 64;
    blk[0] = dcs[b];
    for (int j = 0; j < n; ++j) {
      const int pos = e[j] >> 10;
      if (pos) blk[pos] = (int16_t)((int16_t)(e[j] << 6) >> 6);         // 10-bit value, sign-extended
      else {
        blk[e[j] & 63] = (int16_t)e[j + 1];
        ++j;
      }
    }
  }
}

static void cache_insert(CoefCache* c, const char* path, const comic_jpeg_info* info, const uint16_t* pk, int64_t n_u16) {
  if (!c || !c->buckets || c->bytes >= c->max_bytes) return;
  const int64_t cost = n_u16 * 2 + (int64_t)sizeof(CacheEntry) + (int64_t)strlen(path) + 1;
  CacheEntry* e = (CacheEntry*)calloc(1, sizeof(CacheEntry));
  uint16_t* packed = (uint16_t*)malloc((size_t)n_u16 * 2);
  char* p = strdup(path);
  if (!e || !packed || !p) {
    free(e); free(packed); free(p);
    return;
  }
  memcpy(packed, pk, (size_t)n_u16 * 2);
  e->path = p;
  e->hash = path_hash(path);
  e->info = *info;
  e->info.coef_base = e->info.pixel_off = 0;
  e->packed = packed;
  e->n_u16 = n_u16;
  if (file_stamp(path, &e->file_size, &e->mtime_ns)) {      // (gone since it was read: nothing to key the entry by)
    free(e->path); free(e->packed); free(e);
    return;
  }
  pthread_rwlock_wrlock(&c->lock);
  CacheEntry** slot = &c->buckets[e->hash % (uint64_t)c->n_buckets];
  const CacheEntry* dup = *slot;
  while (dup && !(dup->hash == e->hash && !strcmp(dup->path, path))) dup = dup->next;
  // An entry of this path with ANOTHER stamp is the decode of a file that has been rewritten since: cache_lookup refuses it
  // from now on, so the fresh decode goes in FRONT of it (a lookup stops at the first entry of a path) instead of being
  // dropped as a duplicate -- otherwise a file rewritten once was decoded again in every later epoch.  The stale entry stays
  // allocated and counted: entries are never freed while the pool lives (a reader may still be copying from one).
  const int stale = dup && (dup->file_size != e->file_size || dup->mtime_ns != e->mtime_ns);
  const int take = (!dup || stale) && c->bytes + cost <= c->max_bytes;
  if (take) {
    e->next = *slot;
    *slot = e;
    c->bytes += cost;
    c->entries += 1;
  }
  pthread_rwlock_unlock(&c->lock);
  if (!take) {
    free(e->path); free(e->packed); free(e);
  }
}

static void cache_free(CoefCache* c) {
  if (!c->buckets) return;
  for (int64_t i = 0; i < c->n_buckets; ++i)
    for (CacheEntry* e = c->buckets[i]; e;) {
      CacheEntry* nx = e->next;
      free(e->path); free(e->packed); free(e);
      e = nx;
    }
  free(c->buckets);
  c->buckets = NULL;
  pthread_rwlock_destroy(&c->lock);
}

struct comic_jpeg_pool {
  pthread_mutex_t mu;
  pthread_cond_t work, done;
  Batch* head;                     // batches with passes left, in submission order
  Batch* fin;                      // finished batches whose waiter has not come yet (freed by the waiter, or by destroy)
  int stop, nthreads;
  pthread_t* threads;
  CoefCache cache;
};

static int read_file(const char* path, uint8_t** out, int64_t* out_n) {
  const int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return COMIC_JPEG_IO;
  struct stat st;
  if (fstat(fd, &st) || st.st_size <= 0) {
    close(fd);
    return COMIC_JPEG_IO;
  }
  const size_t n = (size_t)st.st_size;
  uint8_t* data = (uint8_t*)malloc(n + 8);
  if (!data) {
    close(fd);
    return COMIC_JPEG_IO;
  }
  size_t got = 0;
  while (got < n) {
    const ssize_t r = read(fd, data + got, n - got);
    if (r <= 0) break;
    got += (size_t)r;
  }
  close(fd);
  if (got != n) {
    free(data);
    return COMIC_JPEG_IO;
  }
  memset(data + n, 0, 8);
  *out = data;
  *out_n = (int64_t)n;
  return COMIC_JPEG_OK;
}

static void item_free(Item* it) {
  free(it->data);
  free(it->ps);
  it->data = NULL;
  it->ps = NULL;
}

static void pass1(Batch* b, int i) {
  Item* it = &b->items[i];
  comic_jpeg_info* in = &b->infos[i];
  memset(in, 0, sizeof(*in));
  it->hit = cache_lookup(&b->pool->cache, b->paths[i]);
  if (it->hit) {
    *in = it->hit->info;
    b->status[i] = COMIC_JPEG_OK;
    return;
  }
  int rc = read_file(b->paths[i], &it->data, &it->n);
  if (rc == COMIC_JPEG_OK) {
    it->ps = (Parsed*)malloc(sizeof(Parsed));
    rc = it->ps ? parse(it->data, it->n, it->ps, 1) : COMIC_JPEG_IO;
    if (it->ps) *in = it->ps->info;
  }
  b->status[i] = rc;
  if (rc != COMIC_JPEG_OK) item_free(it);
}

// (under the pool's lock, by the thread that finished the batch's last header)
static void lay_out(Batch* b) {
  int64_t used = 0;
  for (int i = 0; i < b->n; ++i) {
    if (b->status[i] != COMIC_JPEG_OK) continue;
    const int64_t count = b->infos[i].coef_count;            // whole blocks: a multiple of 64 elements
    if (used + count > b->capacity) {
      b->status[i] = COMIC_JPEG_TOO_SMALL;
      item_free(&b->items[i]);
      continue;
    }
    b->infos[i].coef_base = used;
    used += count;
  }
  b->used = used;
}

static void pass2(Batch* b, int i) {
  if (b->status[i] != COMIC_JPEG_OK) return;
  Item* it = &b->items[i];
  int16_t* dst = b->coef + b->infos[i].coef_base;
  if (it->hit) {
    packed_to_dense(it->hit->packed, it->hit->info.coef_count / 64, dst);
    return;
  }
  it->ps->info.coef_base = b->infos[i].coef_base;
  const int rc = decode_scan(it->data, it->n, it->ps, dst);
  b->status[i] = rc;
  CoefCache* c = &b->pool->cache;
  if (rc == COMIC_JPEG_OK && c->buckets && c->bytes < c->max_bytes) {
    const int64_t blocks = b->infos[i].coef_count / 64, n_u16 = dense_to_packed(dst, blocks, NULL);
    uint16_t* pk = (uint16_t*)malloc((size_t)n_u16 * 2);
    if (pk) {
      dense_to_packed(dst, blocks, pk);
      cache_insert(c, b->paths[i], &b->infos[i], pk, n_u16);
      free(pk);
    }
  }
  item_free(it);
}

// ---- packed batches (comic_jpeg_pool_submit_packed): one pass per image ---------------------------------------------------------
// The image is decoded into the calling thread's scratch as [desc | dc | entries], room for it is reserved in the batch's blob
// with one atomic add (the images lie in the order their decodes end; infos[i].pixel_off says where), and copied in.
static __thread uint16_t* tl_pk = NULL;
static __thread int64_t tl_pk_cap = 0;
static __thread Parsed* tl_ps = NULL;

static void pass_packed(Batch* b, int i) {
  comic_jpeg_info* in = &b->infos[i];
  memset(in, 0, sizeof(*in));
  const uint16_t* src = NULL;
  int64_t n_u16 = 0;
  const CacheEntry* hit = cache_lookup(&b->pool->cache, b->paths[i]);
  if (hit) {
    *in = hit->info;
    src = hit->packed;
    n_u16 = hit->n_u16;
  } else {
    uint8_t* data = NULL;
    int64_t n = 0;
    int rc = read_file(b->paths[i], &data, &n);
    if (rc == COMIC_JPEG_OK && !tl_ps) tl_ps = (Parsed*)malloc(sizeof(Parsed));
    if (rc == COMIC_JPEG_OK) rc = tl_ps ? parse(data, n, tl_ps, 1) : COMIC_JPEG_IO;
    if (rc == COMIC_JPEG_OK || rc == COMIC_JPEG_UNSUPPORTED) *in = tl_ps->info;
    if (rc == COMIC_JPEG_OK) {
      const int64_t blocks = in->coef_count / 64, need = blocks * (3 + 126) + 2;
      if (blocks * 126 >= (1 << 25)) rc = COMIC_JPEG_UNSUPPORTED;         // the block descriptors hold 25 bits of entry index
      else if (tl_pk_cap < need) {
        free(tl_pk);
        tl_pk = (uint16_t*)malloc((size_t)need * 2);
        tl_pk_cap = tl_pk ? need : 0;
        if (!tl_pk) rc = COMIC_JPEG_IO;
      }
      if (rc == COMIC_JPEG_OK) {
        int64_t cnt = 0;
        rc = decode_scan_packed(data, n, tl_ps, (uint32_t*)tl_pk, (int16_t*)(tl_pk + 2 * blocks), tl_pk + 3 * blocks, &cnt);
        src = tl_pk;
        n_u16 = 3 * blocks + cnt;
        if (n_u16 & 1) tl_pk[n_u16++] = 0;                                 // whole 32-bit words
        if (rc == COMIC_JPEG_OK) cache_insert(&b->pool->cache, b->paths[i], in, src, n_u16);
      }
    }
    free(data);
    if (rc != COMIC_JPEG_OK) {
      b->status[i] = rc;
      return;
    }
  }
  const int64_t off = __atomic_fetch_add(&b->used, n_u16, __ATOMIC_RELAXED);
  if (off + n_u16 > b->capacity) {
    b->status[i] = COMIC_JPEG_TOO_SMALL;
    return;
  }
  memcpy((uint16_t*)b->coef + off, src, (size_t)n_u16 * 2);
  in->pixel_off = off;
  b->status[i] = COMIC_JPEG_OK;
}

static void* pool_worker(void* arg) {
  comic_jpeg_pool* pool = (comic_jpeg_pool*)arg;
  pthread_mutex_lock(&pool->mu);
  for (;;) {
    Batch* b = pool->head;
    int i = -1, pass = 0;
    for (; b; b = b->link) {
      if (b->next1 < b->n) {
        i = b->next1++;
        pass = b->packed ? 3 : 1;
        break;
      }
      if (b->packed) continue;
      if (b->done1 == b->n && b->next2 < b->n) {
        i = b->next2++;
        pass = 2;
        break;
      }
    }
    if (!pass) {
      if (pool->stop && !pool->head) break;
      pthread_cond_wait(&pool->work, &pool->mu);
      continue;
    }
    pthread_mutex_unlock(&pool->mu);
    if (pass == 1) pass1(b, i);
    else if (pass == 2) pass2(b, i);
    else pass_packed(b, i);
    pthread_mutex_lock(&pool->mu);
    if (pass == 3) {
      if (++b->done1 == b->n) {
        b->done2 = b->n;
        Batch** at = &pool->head;
        while (*at && *at != b) at = &(*at)->link;
        if (*at) *at = b->link;
        b->link = pool->fin;
        pool->fin = b;
        pthread_cond_broadcast(&pool->done);
        if (pool->stop) pthread_cond_broadcast(&pool->work);
      }
    } else if (pass == 1) {
      if (++b->done1 == b->n) {
        lay_out(b);
        pthread_cond_broadcast(&pool->work);                 // pass 2 of this batch is open
      }
    } else if (++b->done2 == b->n) {
      Batch** at = &pool->head;                              // out of the work list; the waiter (or destroy) frees it
      while (*at && *at != b) at = &(*at)->link;
      if (*at) *at = b->link;
      b->link = pool->fin;
      pool->fin = b;
      pthread_cond_broadcast(&pool->done);
      if (pool->stop) pthread_cond_broadcast(&pool->work);
    }
  }
  pthread_mutex_unlock(&pool->mu);
  free(tl_pk);
  free(tl_ps);
  tl_pk = NULL;
  tl_ps = NULL;
  tl_pk_cap = 0;
  return NULL;
}

comic_jpeg_pool* comic_jpeg_pool_create(int threads) {
  if (threads < 1 || threads > 1024) return NULL;
  comic_jpeg_pool* pool = (comic_jpeg_pool*)calloc(1, sizeof(*pool));
  if (!pool) return NULL;
  pthread_mutex_init(&pool->mu, NULL);
  pthread_cond_init(&pool->work, NULL);
  pthread_cond_init(&pool->done, NULL);
  pool->threads = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  if (!pool->threads) {
    free(pool);
    return NULL;
  }
  for (int i = 0; i < threads; ++i) {
    if (pthread_create(&pool->threads[i], NULL, pool_worker, pool)) break;
    pool->nthreads++;
  }
  if (!pool->nthreads) {
    free(pool->threads);
    free(pool);
    return NULL;
  }
  return pool;
}

static void batch_free(Batch* b);

void comic_jpeg_pool_destroy(comic_jpeg_pool* pool) {
  if (!pool) return;
  pthread_mutex_lock(&pool->mu);
  pool->stop = 1;
  pthread_cond_broadcast(&pool->work);
  pthread_mutex_unlock(&pool->mu);
  for (int i = 0; i < pool->nthreads; ++i) pthread_join(pool->threads[i], NULL);
  // the workers drained the queue before they left: what is still listed are finished batches nobody waited for (a stage
  // closed with batches in flight, or a wait that timed out)
  while (pool->fin) {
    Batch* b = pool->fin;
    pool->fin = b->link;
    batch_free(b);
  }
  pthread_mutex_destroy(&pool->mu);
  pthread_cond_destroy(&pool->work);
  pthread_cond_destroy(&pool->done);
  cache_free(&pool->cache);
  free(pool->threads);
  free(pool);
}

int comic_jpeg_pool_enable_cache(comic_jpeg_pool* pool, int64_t max_bytes) {
  if (!pool || max_bytes < 0) return COMIC_JPEG_CORRUPT;
  CoefCache* c = &pool->cache;
  if (c->buckets) {                                      // already on: only the limit moves
    pthread_rwlock_wrlock(&c->lock);
    c->max_bytes = max_bytes;
    pthread_rwlock_unlock(&c->lock);
    return COMIC_JPEG_OK;
  }
  if (max_bytes == 0) return COMIC_JPEG_OK;
  pthread_mutex_lock(&pool->mu);
  const int busy = pool->head != NULL;       // workers read c->buckets unlocked: the table must exist before the first submit
  pthread_mutex_unlock(&pool->mu);
  if (busy) return COMIC_JPEG_UNSUPPORTED;
  c->n_buckets = 1 << 18;
  c->buckets = (CacheEntry**)calloc((size_t)c->n_buckets, sizeof(CacheEntry*));
  if (!c->buckets) return COMIC_JPEG_IO;
  pthread_rwlock_init(&c->lock, NULL);
  c->max_bytes = max_bytes;
  return COMIC_JPEG_OK;
}

int comic_jpeg_pool_cache_stats(comic_jpeg_pool* pool, int64_t* bytes, int64_t* entries, int64_t* hits) {
  if (!pool) return COMIC_JPEG_CORRUPT;
  CoefCache* c = &pool->cache;
  if (bytes) *bytes = c->buckets ? c->bytes : 0;
  if (entries) *entries = c->buckets ? c->entries : 0;
  if (hits) *hits = c->buckets ? __atomic_load_n(&c->hits, __ATOMIC_RELAXED) : 0;
  return COMIC_JPEG_OK;
}

static void batch_free(Batch* b) {
  if (!b) return;
  if (b->paths)
    for (int i = 0; i < b->n; ++i) free(b->paths[i]);
  if (b->items)
    for (int i = 0; i < b->n; ++i) item_free(&b->items[i]);
  free(b->paths);
  free(b->items);
  free(b);
}

static void* pool_submit(comic_jpeg_pool* pool, const char* const* paths, int n, comic_jpeg_info* infos, int32_t* status,
                         int16_t* coef, int64_t capacity, int packed);

void* comic_jpeg_pool_submit(comic_jpeg_pool* pool, const char* const* paths, int n, comic_jpeg_info* infos, int32_t* status,
                             int16_t* coef, int64_t capacity) {
  return pool_submit(pool, paths, n, infos, status, coef, capacity, 0);
}

void* comic_jpeg_pool_submit_packed(comic_jpeg_pool* pool, const char* const* paths, int n, comic_jpeg_info* infos,
                                    int32_t* status, uint16_t* packed, int64_t capacity_u16) {
  if (((uintptr_t)packed & 3) != 0) return NULL;                          // the block descriptors are 32-bit words
  return pool_submit(pool, paths, n, infos, status, (int16_t*)packed, capacity_u16, 1);
}

static void* pool_submit(comic_jpeg_pool* pool, const char* const* paths, int n, comic_jpeg_info* infos, int32_t* status,
                         int16_t* coef, int64_t capacity, int packed) {
  if (!pool || !paths || n <= 0 || !infos || !status || !coef || capacity <= 0) return NULL;
  Batch* b = (Batch*)calloc(1, sizeof(Batch));
  if (!b) return NULL;
  b->n = n;
  b->paths = (char**)calloc((size_t)n, sizeof(char*));
  b->items = (Item*)calloc((size_t)n, sizeof(Item));
  if (!b->paths || !b->items) {
    batch_free(b);
    return NULL;
  }
  for (int i = 0; i < n; ++i) {
    b->paths[i] = paths[i] ? strdup(paths[i]) : NULL;
    if (!b->paths[i]) {
      batch_free(b);
      return NULL;
    }
  }
  b->infos = infos;
  b->status = status;
  b->coef = coef;
  b->capacity = capacity;
  b->packed = packed;
  b->pool = pool;
  pthread_mutex_lock(&pool->mu);
  Batch** at = &pool->head;
  while (*at) at = &(*at)->link;
  *at = b;
  pthread_cond_broadcast(&pool->work);
  pthread_mutex_unlock(&pool->mu);
  return b;
}

int comic_jpeg_pool_wait(comic_jpeg_pool* pool, void* batch, double timeout_s, int64_t* coef_elems, int64_t* pixel_bytes) {
  if (!pool || !batch) return COMIC_JPEG_CORRUPT;
  Batch* b = (Batch*)batch;
  struct timespec until;
  clock_gettime(CLOCK_REALTIME, &until);
  if (timeout_s < 0) timeout_s = 0;
  until.tv_sec += (time_t)timeout_s;
  until.tv_nsec += (long)((timeout_s - (double)(time_t)timeout_s) * 1e9);
  if (until.tv_nsec >= 1000000000L) {
    until.tv_sec += 1;
    until.tv_nsec -= 1000000000L;
  }
  pthread_mutex_lock(&pool->mu);
  while (b->done2 < b->n)
    if (pthread_cond_timedwait(&pool->done, &pool->mu, &until)) break;
  const int finished = b->done2 == b->n;
  if (finished) {
    Batch** at = &pool->fin;
    while (*at && *at != b) at = &(*at)->link;
    if (*at) *at = b->link;
  }
  pthread_mutex_unlock(&pool->mu);
  if (!finished) return 1;                                         // stays listed: comic_jpeg_pool_destroy frees it
  int64_t off = 0;
  if (b->packed) {
    // the component planes of the images back to back (coef_base, in samples = bytes); pixel_off stays the packed offset
    for (int i = 0; i < b->n; ++i)
      if (b->status[i] == COMIC_JPEG_OK) {
        b->infos[i].coef_base = off;
        off += b->infos[i].coef_count;
      }
    if (b->used > b->capacity) b->used = b->capacity;            // (reservations of images that did not fit)
  } else {
    for (int i = 0; i < b->n; ++i)
      if (b->status[i] == COMIC_JPEG_OK) {
        b->infos[i].pixel_off = off;
        off += ((int64_t)b->infos[i].width * b->infos[i].height * 3 + 15) & ~(int64_t)15;
      }
  }
  if (pixel_bytes) *pixel_bytes = off;
  if (coef_elems) *coef_elems = b->used;
  batch_free(b);
  return 0;
}
