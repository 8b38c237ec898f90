// Streaming stem kernel (bf16) for gfx950: Conv2d_2a_3x3 (32 -> 32, VALID) -> Conv2d_2b_3x3 (32 -> 64, SAME) ->
// MaxPool_3a_3x3 (3x3 / 2, VALID) in ONE pass over the image.
//
// Replaces the three TF-1.9 op call-sites common/nets/inception_v3.py:104-111 (slim.conv2d x2 under
// inception_arg_scope, common/nets/inception_utils.py:32-82: Conv2D -> FusedBatchNorm(inference) -> Relu, then
// slim.max_pool2d) for forward-only bf16 plans.  As separate launches these layers are a quarter of the forward at
// 640 images: 109 x 109 maps with 32 / 64 channels are too thin for the im2col / patch tiles (432 / 513 TFLOP/s) and
// the 64-channel map is written (973 MB) and re-read by the pool.
//
// Line-buffer formulation.  A persistent workgroup takes (image, half of the pooled rows) tasks and walks down the
// image one row per step:
//   step t:  waves 4-7  X0 row q+3   global -> registers (two steps ahead) -> LDS ring (4 rows, 96-byte pixels)
//                       Y1 row q     = relu(bn(conv3x3(X0 rows q..q+2)))      -> LDS ring (4 rows, zero pad columns)
//            waves 0-3  Y2 row q-2   = conv3x3(Y1 rows q-3..q-1) (raw products)  -> registers
//                       pooled row   = running max over three consecutive Y2 rows kept in registers, on every second
//                                      row a 3-tap max along the row (DPP lane shifts on the accumulator layout), then
//                                      BatchNorm + ReLU (monotonic: they commute with the maximum) -> global
//   one s_barrier per step.  Every X0 byte is read once, Y1 and Y2 never leave the CU, only the pooled map is written.
// All weights (9 taps x 32 k each) stay in registers: a k-step of 32 is exactly one filter tap, so the pixel operand
// of tap (kh, kw) is the 64 contiguous bytes of pixel (x + kw) in ring row (q + kh); the 96-byte pixel stride makes
// the ds_read_b128 lane groups conflict-free (MI355X_MICROARCH.md, LDS).  Every SIMD hosts one Y2 wave and one Y1
// wave; the Y1 wave issues its MFMAs at raised priority so that its epilogue, ring store and row request run under the
// Y2 wave's MFMAs.  Y2 wave (p, h) owns channels [32p, 32p+32) x four 16-pixel tiles placed 14 pixels apart (every
// pooling window inside one accumulator register), Y1 wave (p, h) channels [16p, 16p+16) x pixels [64h, 64h+64).
// k order per accumulator = taps ascending = the im2col kernels' order: the pooled map is bit-identical to the three
// separate launches (tests/test_gpu_path.py).
//
// FUSE1A (op kind 9): Conv2d_1a_3x3 (3x3 / 2 VALID, 3 -> 32, inception_v3.py:100-104) joins the pass.  The Y1 waves no
// longer load X0 rows: they load the fp32 IMAGE rows (two new rows of 224 x 3 floats per step, 16-byte loads, two steps
// ahead -> a 16-row LDS ring), gather the 27-deep im2col operand of X0 row q+3 from the ring (k = 8 fg + s of the lane,
// the stand-alone stem kernel's lane -> k map), split it into bf16 hi / lo and issue the same hi*hi + hi*lo + lo*hi
// 16x16x32 MFMAs, BatchNorm + ReLU epilogue and bf16 rounding as conv_stem_mfma_kernel -- X0 is bit-identical to the
// separate launch, but the 1.0 GB it wrote and this kernel re-read per 1280 images never exist (the image is 0.77 GB).
#include <algorithm>

#include "conv_common.h"
#include "conv_stem.h"

namespace {

constexpr int kPxB = 96;                 // bytes per pixel in the LDS rings (64 data + 32 pad)
constexpr int kRowPx = 132;              // pixels per ring row (W0 <= 116, + tile overrun of the last fragment reads)
constexpr int kRowB = kRowPx * kPxB;
constexpr int kRing = 4;
constexpr int kTableFloats = 2 * (32 + 64);
constexpr int kImgRing = 16;             // FUSE1A: image rows resident in LDS (fp32, Wi * 3 floats each)

#define STEM_BARRIER()                                    \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("" ::: "memory");                        \
  } while (0)

// lane i <- lane i+1 / i+2 of its 16-lane row (row_ror:15 / row_ror:14: the wrapped lanes are patched by the caller)
__device__ __forceinline__ float row_next1(float v) { return dpp_move<0x12F>(v, v); }
__device__ __forceinline__ float row_next2(float v) { return dpp_move<0x12E>(v, v); }

template <bool FUSE1A>
__global__ __launch_bounds__(512) void conv_stem_stream_kernel(ComicStemArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* table = (float*)smem;                                   // sc1[32] sh1[32] sc2[64] sh2[64]
  unsigned char* xring = smem + kTableFloats * 4;
  unsigned char* yring = xring + kRing * kRowB;
  float* imgring = (float*)(yring + kRing * kRowB);              // FUSE1A: [kImgRing][Wi * 3]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H1 = a.H0 - 2, W1 = a.W0 - 2;
  const int part_rows = (a.Hp + a.parts - 1) / a.parts;      // pooled rows of a task: an image is cut into `parts` bands of rows

  // ---- one-time LDS initialisation: tables, zero pad columns of the Y1 ring ----------------------------------------
  if (tid < 32) {
    table[tid] = a.sc1[tid];
    table[32 + tid] = a.sh1[tid];
  }
  if (tid < 64) {
    table[64 + tid] = a.sc2[tid];
    table[128 + tid] = a.sh2[tid];
  }
  for (int i = tid; i < kRing * kRowPx * (kPxB / 16); i += 512) *(uint4*)(yring + i * 16) = make_uint4(0, 0, 0, 0);
  if constexpr (FUSE1A)   // finite contents from the start: the k >= 27 operand positions meet zero weights
    for (int i = tid; i < kImgRing * a.Wi * 3 / 4; i += 512) *(float4*)(imgring + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);

  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t x_lane = (uint32_t)(fr * kPxB + fg * 16);

  if (wave >= 4) {
    // ------------------------------------------------ Y1 waves: X0 row loads + Conv2d_2a ---------------------------
    // wave (p, hh): channels [16p, 16p+16) of pixels [64*hh, 64*hh + 64) -- always four 16-pixel tiles; tiles past
    // W1 multiply ring garbage and are not written.
    const int wv = wave - 4;
    const int p = wv & 1, hh = wv >> 1;
    const int y1_start = 64 * hh;
    const bool y1_work = y1_start < W1;
    const int ltid = tid - 256;
    int px[2], ch[2];
    bool act[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = ltid + 256 * i;
      px[i] = c >> 2;
      ch[i] = c & 3;
      act[i] = px[i] < a.W0;
    }
    // branch-free (clamped addresses): with the loads behind per-lane branches hipcc cannot count them and drains
    // vmcnt(0) at the top of every step, i.e. a step waits for the row it has just requested
    uint32_t goff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) goff[i] = (uint32_t)(min(px[i], a.W0 - 1) * a.x_cs + ch[i] * 8);
    struct Row { uint4 c0, c1; };
    const uint32_t goff0 = goff[0], goff1 = goff[1];
    auto load_row = [&](int b, int row) {
      const int rc = min(max(row, 0), a.H0 - 1);
      const bf16_t* rp = a.x + ((size_t)(b * a.H0 + rc) * a.W0) * a.x_cs + a.x_co;
      Row v;
      v.c0 = *(const uint4*)(rp + goff0);
      v.c1 = *(const uint4*)(rp + goff1);
      return v;
    };
    const uint32_t soff0 = px[0] * kPxB + ch[0] * 16, soff1 = px[1] * kPxB + ch[1] * 16;
    const bool act0 = act[0], act1 = act[1];
    auto store_row = [&](int row, const Row& v) {
      unsigned char* dst = xring + ((row + 8) & 3) * kRowB;
      if (act0) *(uint4*)(dst + soff0) = v.c0;
      if (act1) *(uint4*)(dst + soff1) = v.c1;
    };
    bf16x8_t w1[9];
    {
      const bf16_t* wp1 = a.w1 + (size_t)(p * 16 + fr) * a.Kpad + fg * 8;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) w1[tap] = __builtin_bit_cast(bf16x8_t, *(const uint4*)(wp1 + tap * 32));
    }
    // ---- FUSE1A: Conv2d_1a from the image ------------------------------------------------------------------------------
    const int rowf = a.Wi * 3, row4 = rowf >> 2;          // floats / 16-byte pieces per image row (Wi % 4 == 0)
    // the two NEW image rows of X0 row j (2j+1, 2j+2: contiguous in memory), two 16-byte pieces a thread
    struct ImgRows { float4 c0, c1; };
    auto x0_wanted = [&](int j) { return j >= 0 && j < a.H0; };
    auto load_img = [&](int b, int j) {
      ImgRows v;
      const int jc = min(max(j, 0), a.H0 - 1);
      const float4* src = (const float4*)(a.img + ((size_t)b * a.Hi + 2 * jc + 1) * rowf);
      v.c0 = src[min(ltid, 2 * row4 - 1)];
      v.c1 = src[min(ltid + 256, 2 * row4 - 1)];
      return v;
    };
    auto store_img = [&](int j, const ImgRows& v) {
      if (!x0_wanted(j)) return;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = ltid + 256 * e;
        if (i < 2 * row4) {
          const int r = 2 * j + 1 + (i >= row4 ? 1 : 0), c4 = i >= row4 ? i - row4 : i;
          *(float4*)(imgring + (size_t)(r & (kImgRing - 1)) * rowf + 4 * c4) = e == 0 ? v.c0 : v.c1;
        }
      }
    };
    // Wave wv of the four owns the pixel tiles 2 wv, 2 wv + 1 (32 pixels) x all 32 channels: the gathered and split pixel
    // operand serves both channel tiles.
    bf16x8_t w0h[2], w0l[2];
    int k_kh[8], k_col[8];                 // lane's eight k = 8 fg + s: filter row, float offset (kw * 3 + c) inside an image row
    float4 sc0[2], sh0[2];
    if constexpr (FUSE1A) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        float wf[8];
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
          const int k = 8 * fg + s8;
          const bool kv = k < 27;
          const int kk = kv ? k : 0;
          k_kh[s8] = kv ? kk / 9 : 0;        // k >= 27: zero weights against finite ring contents (the ring starts zeroed)
          k_col[s8] = kk % 9;
          wf[s8] = kv ? a.w0[kk * 32 + n * 16 + fr] : 0.f;
        }
        uint32_t hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hh[e] = pack_bf16x2(wf[2 * e], wf[2 * e + 1]);
          ll[e] = pack_bf16x2(wf[2 * e] - __uint_as_float(hh[e] << 16), wf[2 * e + 1] - __uint_as_float(hh[e] & 0xFFFF0000u));
        }
        w0h[n] = __builtin_bit_cast(bf16x8_t, make_uint4(hh[0], hh[1], hh[2], hh[3]));
        w0l[n] = __builtin_bit_cast(bf16x8_t, make_uint4(ll[0], ll[1], ll[2], ll[3]));
        sc0[n] = *(const float4*)(a.sc0 + n * 16 + fg * 4);
        sh0[n] = *(const float4*)(a.sh0 + n * 16 + fg * 4);
      }
    }
    // X0 row j = relu(bn(conv3x3/2(image))) for this wave's 32 pixels x 32 channels -> X0 ring (the arithmetic of
    // conv_stem_mfma_kernel<bf16_t, 2>: same operand split, same MFMA order per accumulator, same epilogue)
    auto compute_x0 = [&](int j) {
      if (!x0_wanted(j) || !(32 * wv < a.W0)) return;
      unsigned char* dst = xring + ((j + 8) & 3) * kRowB + (fg * 4) * 2;
      int rb[8];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) rb[s8] = ((2 * j + k_kh[s8]) & (kImgRing - 1)) * rowf + k_col[s8];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int x = 32 * wv + 16 * i + fr;
        const int x6 = 6 * min(x, a.W0 - 1);
        float xv[8];
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) xv[s8] = imgring[rb[s8] + x6];
        uint32_t hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hh[e] = pack_bf16x2(xv[2 * e], xv[2 * e + 1]);
          ll[e] = pack_bf16x2(xv[2 * e] - __uint_as_float(hh[e] << 16), xv[2 * e + 1] - __uint_as_float(hh[e] & 0xFFFF0000u));
        }
        const bf16x8_t xh = __builtin_bit_cast(bf16x8_t, make_uint4(hh[0], hh[1], hh[2], hh[3]));
        const bf16x8_t xl = __builtin_bit_cast(bf16x8_t, make_uint4(ll[0], ll[1], ll[2], ll[3]));
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          f32x4_t acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0h[n], xh, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0h[n], xl, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0l[n], xh, acc0, 0, 0, 0);
          const float v0 = fmaxf(fmaf(acc0[0], sc0[n].x, sh0[n].x), 0.f), v1 = fmaxf(fmaf(acc0[1], sc0[n].y, sh0[n].y), 0.f);
          const float v2 = fmaxf(fmaf(acc0[2], sc0[n].z, sh0[n].z), 0.f), v3 = fmaxf(fmaf(acc0[3], sc0[n].w, sh0[n].w), 0.f);
          if (x < a.W0) *(uint2*)(dst + x * kPxB + n * 32) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        }
      }
    };
    STEM_BARRIER();                                  // LDS initialised
    const float4 sc1 = *(const float4*)(table + p * 16 + fg * 4), sh1 = *(const float4*)(table + 32 + p * 16 + fg * 4);
    for (int task = blockIdx.x; task < a.n_tasks; task += gridDim.x) {
      const int b = task / a.parts, h = task - b * a.parts;
      const int p0 = h * part_rows, p1 = min(a.Hp, p0 + part_rows) - 1;
      const int qa = 2 * p0 - 1, r_last = 2 * p1 + 2;
      const int nsteps = r_last + 2 - qa + 1;
      Row r0, r1, r2;
      ImgRows ia, ib, ic;
      if constexpr (FUSE1A) {
        // image rows of X0 rows qa .. qa+3 (2 qa .. 2 qa + 8, those inside the image) straight into the ring
        const int ir0 = 2 * max(qa, 0), ir1 = min(2 * (qa + 3) + 2, a.Hi - 1);
        const float4* src = (const float4*)(a.img + ((size_t)b * a.Hi + ir0) * rowf);
        for (int i = ltid; i < (ir1 - ir0 + 1) * row4; i += 256) {
          const int r = ir0 + i / row4, c4 = i % row4;
          *(float4*)(imgring + (size_t)(r & (kImgRing - 1)) * rowf + 4 * c4) = src[i];
        }
        ia = load_img(b, qa + 4);                    // queue: stored in step 0 / step 1
        ib = load_img(b, qa + 5);
        STEM_BARRIER();                              // (matched by the Y2 waves) the image rows are in the ring
        compute_x0(qa);
        compute_x0(qa + 1);
        compute_x0(qa + 2);
      } else {
        r0 = load_row(b, qa); r1 = load_row(b, qa + 1); r2 = load_row(b, qa + 2);
        store_row(qa, r0);
        store_row(qa + 1, r1);
        store_row(qa + 2, r2);
        r0 = load_row(b, qa + 3);                    // queue: r0 = the row stored in step 0, r1 = step 1
        r1 = load_row(b, qa + 4);
      }
      STEM_BARRIER();                                // rows qa .. qa+2 are in the ring
      for (int t = 0; t < nsteps; ++t) {
        const int q = qa + t;
        if constexpr (FUSE1A) ic = load_img(b, q + 6);
        else r2 = load_row(b, q + 5);
        // ---- Y1 row q (rows -1 and H1 are the zero padding of the SAME conv that follows) -------------------------
        if (q <= r_last + 1 && y1_work) {
          const bool real = q >= 0 && q < H1;
          f32x4_t acc[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          if (real) {
            bf16x8_t xf[4][4];
            auto frags = [&](int tap, bf16x8_t (&f)[4]) {
              const int kh = tap / 3, kw = tap % 3;
              const unsigned char* rowp = xring + ((q + kh + 8) & 3) * kRowB + (y1_start + kw) * kPxB + x_lane;
#pragma unroll
              for (int j = 0; j < 4; ++j) f[j] = __builtin_bit_cast(bf16x8_t, *(const uint4*)(rowp + j * 16 * kPxB));
            };
            // fragment reads run three taps ahead of the MFMAs (four MFMAs = 64 cycles do not cover an LDS round trip)
            frags(0, xf[0]);
            frags(1, xf[1]);
            frags(2, xf[2]);
            // this wave's 36 MFMAs go first on the SIMD's matrix pipe: its epilogue, ring store and row request then run
            // under the 72 MFMAs of the Y2 wave it shares the SIMD with (at equal priority both finish together and
            // neither tail is covered)
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
              if (tap + 3 < 9) frags(tap + 3, xf[(tap + 3) & 3]);
#pragma unroll
              for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[tap], xf[tap & 3][j], acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
          }
          // lane holds channels 16p + 4fg .. +3 of pixel y1_start + 16j + fr -> ring column (pixel + 1)
          unsigned char* dst = yring + ((q + 8) & 3) * kRowB + (p * 16 + fg * 4) * 2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int x = y1_start + j * 16 + fr;
            float v0 = fmaxf(fmaf(acc[j][0], sc1.x, sh1.x), 0.f), v1 = fmaxf(fmaf(acc[j][1], sc1.y, sh1.y), 0.f);
            float v2 = fmaxf(fmaf(acc[j][2], sc1.z, sh1.z), 0.f), v3 = fmaxf(fmaf(acc[j][3], sc1.w, sh1.w), 0.f);
            if (!real) v0 = v1 = v2 = v3 = 0.f;
            if (x < W1) *(uint2*)(dst + (x + 1) * kPxB) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
          }
        }
        if constexpr (FUSE1A) {
          compute_x0(q + 3);                         // from image rows stored one and two steps ago
          store_img(q + 4, ia);
          ia = ib;
          ib = ic;
        } else {
          store_row(q + 3, r0);
          r0 = r1;
          r1 = r2;
        }
        STEM_BARRIER();
      }
    }
    return;
  }

  // ---------------------------------------------------- Y2 waves: Conv2d_2b + MaxPool_3a --------------------------
  // wave (p, hh): channels [32p, 32p+32) of four 16-pixel tiles placed 14 pixels apart from pixel 56*hh: the pooling
  // windows [x, x+2] with x = tile origin + 0, 2, .. 12 lie inside ONE tile (7 pooled columns per tile, 28 per wave), so
  // the row maximum needs no value from another accumulator register.
  const int p = wave & 1, hh = wave >> 1;
  const int y2_start = 56 * hh;
  const bool y2_work = y2_start < W1;
  constexpr int kTileStep = 14;
  bf16x8_t w2[2][9];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bf16_t* wp2 = a.w2 + (size_t)((2 * p + i) * 16 + fr) * a.Kpad + fg * 8;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) w2[i][tap] = __builtin_bit_cast(bf16x8_t, *(const uint4*)(wp2 + tap * 32));
  }
  STEM_BARRIER();                                    // LDS initialised
  float4 sc2[2], sh2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    sc2[i] = *(const float4*)(table + 64 + (2 * p + i) * 16 + fg * 4);
    sh2[i] = *(const float4*)(table + 128 + (2 * p + i) * 16 + fg * 4);
  }

  for (int task = blockIdx.x; task < a.n_tasks; task += gridDim.x) {
    const int b = task / a.parts, h = task - b * a.parts;
    const int p0 = h * part_rows, p1 = min(a.Hp, p0 + part_rows) - 1;
    const int r_first = 2 * p0, r_last = 2 * p1 + 2;
    const int qa = r_first - 1;
    const int nsteps = r_last + 2 - qa + 1;
    // running maximum of the RAW products over the rows of the current pooling window.  BatchNorm (scale > 0: no gamma,
    // inception_utils.py:56-66) + ReLU are monotonic, so they commute with the maximum and are applied to the pooled
    // values only.
    f32x4_t run[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) run[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if constexpr (FUSE1A) STEM_BARRIER();             // the Y1 waves' image-row prologue
    STEM_BARRIER();                                   // first three X0 rows are in the ring
    auto mfma_row = [&](int r, f32x4_t (&acc)[2][4]) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      bf16x8_t xf[2][4];
      auto frags = [&](int tap, bf16x8_t (&f)[4]) {
        const int kh = tap / 3, kw = tap % 3;
        // ring column of input pixel (x - 1 + kw) is (x + kw)
        const unsigned char* rowp = yring + ((r - 1 + kh + 8) & 3) * kRowB + (y2_start + kw) * kPxB + x_lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = __builtin_bit_cast(bf16x8_t, *(const uint4*)(rowp + j * kTileStep * kPxB));
      };
      frags(0, xf[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) frags(tap + 1, xf[(tap + 1) & 1]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[i][tap], xf[tap & 1][j], acc[i][j], 0, 0, 0);
      }
    };
    auto pool_row = [&](int r, f32x4_t (&acc)[2][4]) {
      const int rel = r - r_first;
      const bool emit = rel >= 2 && (rel & 1) == 0;
      if (!emit) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) run[i][j][e] = rel == 0 ? acc[i][j][e] : fmaxf(run[i][j][e], acc[i][j][e]);
        return;
      }
      const int prow = p0 + (rel >> 1) - 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[j][e] = fmaxf(run[i][j][e], acc[i][j][e]);      // column maximum over the three rows
            run[i][j][e] = acc[i][j][e];                       // this row opens the next window
          }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)      // pixels x, x+1, x+2 = lanes fr, fr+1, fr+2 of this tile (used for fr <= 12)
            o[e] = fmaxf(fmaxf(v[j][e], row_next1(v[j][e])), row_next2(v[j][e]));
          const float o0 = fmaxf(fmaf(o[0], sc2[i].x, sh2[i].x), 0.f), o1 = fmaxf(fmaf(o[1], sc2[i].y, sh2[i].y), 0.f);
          const float o2 = fmaxf(fmaf(o[2], sc2[i].z, sh2[i].z), 0.f), o3 = fmaxf(fmaf(o[3], sc2[i].w, sh2[i].w), 0.f);
          const int x = y2_start + j * kTileStep + fr;
          const int jc = x >> 1;
          if ((fr & 1) == 0 && fr <= 12 && jc < a.Wp) {
            bf16_t* yp = a.y + ((size_t)((b * a.Hp + prow) * a.Wp + jc)) * a.y_cs + a.y_co + (2 * p + i) * 16 + fg * 4;
            *(uint2*)yp = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
          }
        }
      }
    };
    f32x4_t acc[2][4];
    for (int t = 0; t < nsteps; ++t) {
      const int r = qa + t - 2;
      if (y2_work && r >= r_first && r <= r_last) {
        mfma_row(r, acc);
        pool_row(r, acc);
      }
      STEM_BARRIER();
    }
  }
}

}  // namespace

// Two Y2 waves x four tiles x seven pooled columns = 56 pooled columns; the Y1 halves cover 128 pixels.
bool comic_stem_stream_supported(int H0, int W0) {
  const int Wp = (W0 - 2 - 3) / 2 + 1;
  return H0 >= 7 && W0 >= 7 && Wp <= 56;
}

// FUSE1A: the image rows are moved as 16-byte pieces (Wi % 4 == 0), two new rows by 256 threads with two pieces each, the
// ring fits beside the X0 / Y1 rings, and X0 has the 3x3 / 2 VALID geometry
bool comic_stem_stream_1a_supported(int Hi, int Wi) {
  const int H0 = (Hi - 3) / 2 + 1, W0 = (Wi - 3) / 2 + 1;
  return Hi >= 17 && Wi % 4 == 0 && 2 * (Wi * 3 / 4) <= 512 && comic_stem_stream_supported(H0, W0) &&
         kTableFloats * 4 + 2 * kRing * kRowB + kImgRing * Wi * 3 * 4 <= 160 * 1024;
}

int comic_stem_stream_launch(const ComicStemArgs& a_in, hipStream_t st) {
  ComicStemArgs a = a_in;
  if (!comic_stem_stream_supported(a.H0, a.W0) || (a.img && !comic_stem_stream_1a_supported(a.Hi, a.Wi))) {
    comic_set_error("conv_stem: unsupported map %dx%d", a.H0, a.W0);
    return 2;
  }
  const int lds = kTableFloats * 4 + 2 * kRing * kRowB + (a.img ? kImgRing * a.Wi * 3 * 4 : 0);
  static PerDeviceOnce attr_once__;
  bool& attr_set = attr_once__.slot();   // hipFuncSetAttribute holds per device
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_stem_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv_stem_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      comic_set_error("conv_stem: cannot reserve %d bytes of LDS", lds);
      return 1;
    }
    attr_set = true;
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  // bands of pooled rows per image: two (each band re-computes the five rows above its first pooling window: 59 + 57 row steps
  // for the 111 of an image at 224) while the halves of the batch occupy every CU; small batches take four or eight thinner
  // bands (33 / 20 steps a task) instead of leaving CUs idle behind 59-step tasks (64 images: 126 -> 7x us).  Every output row
  // is produced by the same instructions on the same operands whatever the cut: bit-identical.
  a.parts = 2;
  while (a.parts < 8 && a.B * a.parts < cus && (a.Hp + 2 * a.parts - 1) / (2 * a.parts) >= 6) a.parts *= 2;
  a.n_tasks = a.B * a.parts;
  const int grid = std::min(a.n_tasks, cus);
  if (a.img)
    hipLaunchKernelGGL(conv_stem_stream_kernel<true>, dim3(grid), dim3(512), lds, st, a);
  else
    hipLaunchKernelGGL(conv_stem_stream_kernel<false>, dim3(grid), dim3(512), lds, st, a);
  return 0;
}
