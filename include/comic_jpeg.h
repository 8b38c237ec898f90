/* comic_jpeg.h -- host half of the split JPEG decoder of the input pipeline.
 *
 * The reference decodes every training image inside its tf.data map (tf.image.decode_jpeg, i.e. libjpeg, in
 * common/inputs/manager_image_caption.py:163-175 -> inception_preprocessing_radix.py).  Here the host only
 * undoes the ENTROPY CODING of a baseline JPEG (Huffman codes -> quantised DCT coefficients, this header,
 * libcomic_jpeg.so: plain C, no GPU runtime, callable from loader threads with the interpreter lock released);
 * dequantisation, the inverse DCT, chroma upsampling and the YCbCr -> RGB conversion run on the device
 * (comic_jpeg_pixels in comic_hip.h) with libjpeg's integer arithmetic (jidctint.c "ISLOW", jdsample.c fancy
 * upsampling, jdcolor.c), so the pixels are the bits PIL / libjpeg-turbo produce.
 *
 * Files the split decoder does not take (progressive or arithmetic coding, multi-scan, CMYK / RGB colour spaces,
 * 12-bit samples, sampling other than 1x1 / 2x1 / 2x2 with 1x1 chroma) are reported as COMIC_JPEG_UNSUPPORTED: the
 * loader sends them through its PIL path.
 */
#ifndef COMIC_JPEG_H
#define COMIC_JPEG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define COMIC_JPEG_OK 0
#define COMIC_JPEG_UNSUPPORTED 1   /* a valid file the split decoder does not handle: fall back */
#define COMIC_JPEG_CORRUPT (-1)    /* malformed or truncated stream */
#define COMIC_JPEG_TOO_SMALL (-2)  /* the coefficient buffer handed in is smaller than coef_count */
#define COMIC_JPEG_IO (-3)         /* the file could not be read */

/* Geometry of one image and where its coefficients live.  Identical layout on the device (comic_jpeg_pixels reads an
 * array of these records): 8-byte aligned, 512 bytes. */
typedef struct comic_jpeg_info {
  int32_t width, height;        /* image size in pixels */
  int32_t ncomp;                /* 1 (greyscale) or 3 (YCbCr) */
  int32_t hmax, vmax;           /* luma sampling factors (chroma is 1x1): 1x1, 2x1 or 2x2 */
  int32_t mcus_x, mcus_y;       /* MCU grid */
  int32_t restart_interval;     /* MCUs between RSTn markers (0: none) */
  int32_t blocks_w[3], blocks_h[3];   /* block grid of each component, whole MCUs */
  int32_t comp_w[3], comp_h[3];       /* libjpeg's downsampled_width / _height: the real samples of each component */
  int64_t coef_off[3];          /* first coefficient of each component plane, in int16 elements from the image's base */
  int64_t coef_count;           /* int16 elements of the image: sum of blocks_w * blocks_h * 64 */
  int64_t coef_base;            /* offset of this image in the batch: of its dense coefficients (int16 elements) == of its
                                 * component planes (samples); dense batches: filled by the pool / the caller; packed batches:
                                 * filled by comic_jpeg_pool_wait */
  int64_t pixel_off;            /* dense batches: byte offset of the image's RGB pixels in the batch's pixel blob (comic_jpeg_pixels);
                                 * packed batches: offset of the image's packed coefficients in the blob (16-bit units) */
  uint16_t quant[3][64];        /* quantisation table of each component, natural (row-major) order */
} comic_jpeg_info;

/* Header only (SOI .. SOS): fills `info` (coef_base / pixel_off left 0).  COMIC_JPEG_OK / _UNSUPPORTED / _CORRUPT. */
int comic_jpeg_read_header(const uint8_t* data, int64_t n, comic_jpeg_info* info);

/* Entropy decode of the whole scan: block (by, bx) of component c lands at coef[coef_off[c] + (by * blocks_w[c] + bx) * 64],
 * 64 quantised coefficients in natural order (DC prediction undone, not dequantised).  `coef` must hold
 * info->coef_count elements; every block is written (zeros included). */
int comic_jpeg_decode_coefficients(const uint8_t* data, int64_t n, const comic_jpeg_info* info, int16_t* coef);

/* read_header + decode_coefficients of a file.  `coef_capacity` in int16 elements; COMIC_JPEG_TOO_SMALL leaves `info`
 * filled so that the caller can size a larger buffer. */
int comic_jpeg_decode_file(const char* path, comic_jpeg_info* info, int16_t* coef, int64_t coef_capacity);

/* ---- a batch at a time: a persistent pool of decode threads ------------------------------------------------------------
 * The loader's producer thread queues the files of a batch and goes on (several batches may be queued: the threads flow from
 * one into the next, no batch waits for the stragglers of the one before); no interpreter work per image.
 * Two passes: every file is read and its headers parsed, then the images are laid out BACK TO BACK in `coef` in path order
 * (infos[i].coef_base, whole blocks: multiples of 64 elements) and their scans decoded -- one host-to-device copy of
 * coef[0 .. coef_elems) moves the batch.  status[i] gets the image's code; an image that is unsupported, corrupt, unreadable
 * or does not fit the rest of `capacity` (COMIC_JPEG_TOO_SMALL) takes no room and goes through the loader's PIL path. */
typedef struct comic_jpeg_pool comic_jpeg_pool;
comic_jpeg_pool* comic_jpeg_pool_create(int threads);
void comic_jpeg_pool_destroy(comic_jpeg_pool* pool);          /* waits for queued work; releases the handles nobody waited for */
/* Returns a batch handle (NULL: bad arguments / out of memory).  `paths` is copied; infos / status / coef (capacity int16
 * elements) must stay valid until comic_jpeg_pool_wait has returned 0 for the handle, or comic_jpeg_pool_destroy has returned. */
void* comic_jpeg_pool_submit(comic_jpeg_pool* pool, const char* const* paths, int n, comic_jpeg_info* infos, int32_t* status,
                             int16_t* coef, int64_t capacity);
/* 0: every image of the batch is done -- *coef_elems = elements of `coef` in use; pixel_off of the decoded images assigned
 * back to back (16-byte aligned, width * height * 3 bytes each), their sum in *pixel_bytes; the handle is released.
 * 1: not done within timeout_s (the handle stays valid).  < 0: bad arguments. */
int comic_jpeg_pool_wait(comic_jpeg_pool* pool, void* batch, double timeout_s, int64_t* coef_elems, int64_t* pixel_bytes);

/* The loader's form: PACKED batches.  Image i is decoded to 16-bit units [desc: one uint32 per block][dc: one int16 per block]
 * [entries]: desc[g] = (first entry << 7) | number of entries of block g (blocks in plane order: coef_off[c] / 64 + row *
 * blocks_w[c] + column; first-entry indices count from the image's first entry), dc[g] = the block's DC coefficient, an entry =
 * (natural position 1..63 << 10) | (value & 1023) for a non-zero AC coefficient within -512 .. 511, or the pair (position,
 * value as int16) for a larger one.  3-4x fewer bytes than the dense blocks for the host's memory and the bus (0.28 instead of
 * 0.92 MB for a detailed 640 x 480 image at quality 90), no clearing and no scattered stores on the host; the device entry point
 * of comic_hip.h expands a block in LDS in front of its inverse DCT.  One pass per image; the images lie in `packed` (4-byte
 * aligned) in the order their decodes end: infos[i].pixel_off = offset of image i (16-bit units, even), and after the wait
 * infos[i].coef_base = offset of its component planes (samples) when the planes of the batch's images lie back to back.
 * comic_jpeg_pool_wait returns the 16-bit units in use in *coef_elems and the samples of all planes in *pixel_bytes.  Images of
 * more than 266 000 blocks are COMIC_JPEG_UNSUPPORTED. */
void* comic_jpeg_pool_submit_packed(comic_jpeg_pool* pool, const char* const* paths, int n, comic_jpeg_info* infos,
                                    int32_t* status, uint16_t* packed, int64_t capacity_u16);

/* Coefficient cache of a pool (off by default): every image a batch has decoded is kept -- as its non-zero coefficients, about
 * the size of the JPEG file -- under its path until `max_bytes` are in use (nothing is evicted; call before the first submit -- switching it ON while batches are
 * queued is refused with COMIC_JPEG_UNSUPPORTED -- or again to move the limit).  A later batch that names the path again gets the coefficients from memory: no file read, no Huffman
 * decoding -- the epochs after the first run at the speed of a memory copy.  An entry is served only while the file's size and
 * modification time are those of the decode that made it (one stat per hit); a file rewritten since is decoded from disk
 * (its old entry stays: nothing is evicted).
 * Augmentation (flip, crop) happens on the device, behind the decode, so cached training batches are the bits of uncached ones. */
int comic_jpeg_pool_enable_cache(comic_jpeg_pool* pool, int64_t max_bytes);
int comic_jpeg_pool_cache_stats(comic_jpeg_pool* pool, int64_t* bytes, int64_t* entries, int64_t* hits);

#ifdef __cplusplus
}
#endif
#endif
